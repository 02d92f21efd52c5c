/*
 * mpeg_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the reference decoder gen2brain/mpeg for the
 * hot path and the serial parse in front of it.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or load this; the
 * product library (libmpeghip / libmpeghost) never does.
 *
 * Parity status: PINNED.  The restatement reproduces the reference's own
 * golden hashes (mpeg_test.go:193-197, :227) — see tests/test_oracle_golden.py.
 * Exception: orc_ycbcr_to_rgba restates Go's standard library (image/draw,
 * not vendored in the reference) and no reference test pins it: "parity
 * unpinned" for that one function.
 *
 * Every function cites the reference file:line (relative to the reference
 * repository root) that it follows.
 */
#ifndef MPEG_ORACLE_H
#define MPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------ hot-path units */

/* video.go:801-928 — two-variant integer IDCT on premultiplied coefficients
 * (Go `int` = int64).  max_index is decodeBlock's `n`. Returns the largest
 * |intermediate| seen (guard for the device's int32 arithmetic). */
int64_t orc_idct(int64_t block[64], int max_index);

/* video.go:943-1002 */
void orc_copy_block_to_dest(const int64_t block[64], uint8_t *dest, int index, int scan);
void orc_add_block_to_dest(const int64_t block[64], uint8_t *dest, int index, int scan);
void orc_copy_value_to_dest(int64_t value, uint8_t *dest, int index, int scan);
void orc_add_value_to_dest(int64_t value, uint8_t *dest, int index, int scan);

/* A frame in the reference's layout (video.go:333-355): one allocation
 * Y | Cb | Cr | pad(luma_w*16). */
typedef struct orc_frame {
    uint8_t *base;
    size_t   total;       /* len(base)                                   */
    uint8_t *y, *cb, *cr; /* plane starts inside base                    */
    size_t   luma_size, chroma_size;
    int      luma_w, luma_h, chroma_w, chroma_h;
    int      width, height;
    double   time;
} orc_frame;

int  orc_frame_alloc(orc_frame *f, int width, int height);
void orc_frame_free(orc_frame *f);

/* video_noasm.go:28-43 (+copyBlock :48-80, roundAvg :14-16, bilinAvg :21-26):
 * SWAR motion compensation of one macroblock, all three planes.
 * Returns 0, or -1 (nothing copied) if a read would leave [plane start, end of
 * base) — where Go panics. */
int orc_copy_macroblock(int motion_h, int motion_v, int mb_row, int mb_col,
                        const orc_frame *s, orc_frame *d);
/* video_test.go:10-43 — scalar per-pixel known-answer generator. */
int orc_copy_macroblock_ref(int motion_h, int motion_v, int mb_row, int mb_col,
                            const orc_frame *s, orc_frame *d);
/* video_test.go:45-59 — newTestFrame fill pattern on a square test frame. */
void orc_test_frame_fill(orc_frame *f, int fill);

/* audio.go:492-772 — "idct36", the 32-point matrixing DCT.  s is the
 * reference's sample[ch] = [32][3]int, ss the column, d the V ring, dp=vPos. */
void orc_idct36(const int64_t s[32][3], int ss, float d[1024], int dp);
/* audio_noasm.go:8-38 — polyphase window; fma!=0 contracts each tap to fmaf
 * (what audio_amd64.s:94-97 VFMADD231PS does). */
void orc_synth_window(float u[32], const float d[1024], const float v[1024], int vpos, int fma);
/* audio_test.go:9-31 */
void orc_synth_window_ref(float u[32], const float d[1024], const float v[1024], int vpos, int fma);
/* audio.go:95-98 — window table duplicated into d[0:512] and d[512:1024]. */
void orc_window_table(float d[1024]);

/* Go stdlib image/draw -> image/internal/imageutil.DrawYCbCr, 4:2:0 case, as
 * called by Frame.RGBA (video.go:31-36).  PARITY UNPINNED (see header). */
void orc_ycbcr_to_rgba(const orc_frame *f, uint8_t *rgba /* width*height*4 */);

/* mpeg_test.go:183-186/221-223 — FNV-1a-64. */
#define ORC_FNV_OFFSET 0xcbf29ce484222325ull
uint64_t orc_fnv1a64(uint64_t h, const void *data, size_t n);

/* ------------------------------------------------------------- full decoders */

typedef struct orc_video orc_video;
typedef struct orc_video_stats {
    int pictures[4];          /* decoded pictures by type (index 1=I,2=P,3=B)      */
    int frames_returned;
    int invalid_blocks;       /* early returns at video.go:713                      */
    int coded_blocks, dc_only_blocks, sparse_idct, full_idct;
    int coded_mbs, intra_mbs, skipped_mbs, bidir_mbs;
    int copy_mb_calls, copy_mode[4]; /* 0 full-pel, 1 H, 2 V, 3 HV (luma)          */
    int overreads;            /* MC block reads that end beyond their plane         */
    int range_errors;         /* MC reads outside base (Go would panic)             */
    int64_t max_idct_in, max_idct_out, max_idct_mid;
    int max_abs_mv;
} orc_video_stats;

/* NewVideo (video.go:110-121) over a complete elementary stream in memory. */
orc_video *orc_video_open(const uint8_t *data, size_t len);
void       orc_video_close(orc_video *v);
int    orc_video_has_header(orc_video *v);
int    orc_video_width(orc_video *v);
int    orc_video_height(orc_video *v);
double orc_video_framerate(orc_video *v);
void   orc_video_set_no_delay(orc_video *v, int no_delay);
/* Video.Decode (video.go:209-268): NULL at end. The frame aliases decoder storage. */
const orc_frame *orc_video_decode(orc_video *v);
/* Video.Rewind / Time / HasEnded (video.go:195-201, 183, 203) */
void   orc_video_rewind(orc_video *v);
double orc_video_time(const orc_video *v);
int    orc_video_has_ended(const orc_video *v);
const orc_video_stats *orc_video_get_stats(const orc_video *v);

typedef struct orc_audio orc_audio;
/* NewAudio (audio.go:83-104). fma: 0 none (amd64 Go/SSE2), 1 window only (AVX2). */
orc_audio *orc_audio_open(const uint8_t *data, size_t len, int fma);
void       orc_audio_close(orc_audio *a);
int orc_audio_samplerate(orc_audio *a);
int orc_audio_channels(orc_audio *a);
/* Audio.Decode (audio.go:163-182), AudioF32N: returns Samples.Interleaved
 * (2304 floats) or NULL. `samples_out`, if not NULL, receives the frame's
 * requantised sub-band samples as int32 [2][36][32] (the device input layout). */
const float *orc_audio_decode(orc_audio *a, int32_t *samples_out);
/* Audio.Rewind / Time / HasEnded (audio.go:149-154, 137, 157) */
void   orc_audio_rewind(orc_audio *a);
double orc_audio_time(const orc_audio *a);
int    orc_audio_has_ended(const orc_audio *a);
void orc_audio_get_state(const orc_audio *a, float v[2][1024], int *vpos);

/* demux.go:473-584 subset — pull the payload of every PES packet of `type`
 * (0xE0 video, 0xC0 audio) out of a program stream.  Returns a malloc'd buffer. */
uint8_t *orc_ps_extract(const uint8_t *data, size_t len, int type, size_t *out_len, int *n_packets);

#ifdef __cplusplus
}
#endif
#endif
