/*
 * mpeghost.h — flat C API of libmpeghost, the host-side mirror of gen2brain/mpeg's Go API
 * (bitstream parse on the CPU, reconstruction on the MI355X through libmpeghip, see mpeghip.h).
 *
 * The C++ classes of mpeg_amd/host/mpeg.hpp carry the reference's names (mpeg.MPEG, Video, Audio,
 * Demux, Buffer: mpeg.go, video.go, audio.go, demux.go, buffer.go); this header is the same surface
 * for callers that cannot include C++ — ctypes tests (tests/hostlib.py), cgo, other FFIs.
 *
 * Conventions: handles are opaque void *; functions that can fail return NULL / 0 / -1 and leave a
 * message for the calling thread in mpeghost_last_error(); nothing throws across this boundary.
 * Pointers returned by decode calls alias decoder-owned storage and stay valid until the next decode
 * call on the same handle (as in the reference, mpeg.go:413-415, 435-437).
 */
#ifndef MPEGHOST_H
#define MPEGHOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frame (video.go:11-24): planes are host memory filled on demand from the device. */
typedef struct mpeghost_frame {
    double time;
    int width, height;
    int luma_w, luma_h, chroma_w, chroma_h;
    const uint8_t *y, *cb, *cr;
    size_t luma_bytes, chroma_bytes;
} mpeghost_frame;

const char *mpeghost_last_error(void);

/* one GPU: a libmpeghip context (mpeghip_ctx_create) + what the decoders share on it */
void *mpeghost_device_create(int ordinal);
void mpeghost_device_destroy(void *device);

/* ---- Video (video.go:110 NewVideo over a complete elementary stream, :209 Decode) */
void *mpeghost_video_open(void *device, const uint8_t *data, size_t len);
void *mpeghost_video_open_backend(void *backend, const uint8_t *data, size_t len); /* test backends; takes ownership */
void mpeghost_video_close(void *video);
int mpeghost_video_width(void *video);
int mpeghost_video_height(void *video);
double mpeghost_video_framerate(void *video);
void mpeghost_video_set_no_delay(void *video, int no_delay);        /* video.go:178 */
/* The form the parser hands pictures to libmpeghip in (include/mpeghip.h): 1 = sparse, its own (position, level) pairs
 * (MPEGHIP_PIC_SPARSE; the default), 0 = 128-byte units.  Per decoder (before its first picture), or as the process-wide
 * default of decoders created afterwards (those inside mpeghost_mpeg_* / mpeghost_video_batch_* included).  No reference
 * counterpart: results are identical. */
void mpeghost_video_set_sparse(void *video, int sparse);
void mpeghost_set_default_sparse(int sparse);
/* Test hook: every VLC table of the parser (two-level lookups, mpeg_amd/host/vlc.hpp) against a walk over its ISO 11172-2
 * code list, for all 2^L looks at the stream; returns the number that decode differently (0). */
uint64_t mpeghost_debug_vlc_self_check(void);
/* Test hook: ONE look at the stream through one of the parser's tables — table 0 .. 8 = macroblock_address_increment,
 * macroblock_type I / P / B, coded_block_pattern, motion code, dct_dc_size luma / chroma, the coefficient codes (video.go:1088-1419
 * in that order); `window` = the next 64 bits of the stream, first bit on top.  Yields what buffer.go:352-376's tree walk would
 * return and how many bits it would consume (tests/test_vlc_known_answers.py: every path of the reference's trees).  -1: no such table. */
int mpeghost_debug_vlc_decode(int table, uint64_t window, int *value, int *len);
int mpeghost_video_decode(void *video, mpeghost_frame *out);         /* 1 frame, 0 none / end, -1 error */
const uint8_t *mpeghost_video_rgba(void *video);                     /* Frame.RGBA() of the last decoded frame */
double mpeghost_video_time(void *video);                             /* video.go:183: the time of the next frame */
int mpeghost_video_has_ended(void *video);                           /* video.go:203 */
void mpeghost_video_rewind(void *video);                             /* video.go:195 */
/* Video.Decode works one picture ahead on the host (parse of picture N+1 while picture N is on the device; that picture's work
 * is held back until the next call, so Rewind / Seek see the reference's state).  0 switches it off: parse, submit, read back. */
void mpeghost_video_set_lookahead(void *video, int on);
/* The frames Video.Decode returns come out of the device store's HOST MIRROR (mpeghip_video_host_mirror: the reconstruction launch
 * writes every frame once more, linearly, into pinned host memory — no read-back is queued; the default).  0: the asynchronous
 * read-back into two pinned frames (mpeghip_video_read_planes_async).  Same frames either way; switching ends the life of the frame
 * in hand. */
void mpeghost_video_set_host_mirror(void *video, int on);
/* A lone decoder's hand-overs of n_mbs macroblocks and more (sparse form) are validated and packed by the DEVICE — a device-packed
 * stage of one picture, include/mpeghip.h — instead of by the thread that parses; 0 = never.  Default: 3 000, where the two ways cost the same
 * (a 1080p picture has 8 160 macroblocks, a SIF picture 330).  Same frames either way; what the device would refuse the parser does not hand over. */
void mpeghost_video_set_device_pack_from(void *video, uint32_t n_mbs);
void mpeghost_video_stats(void *video, uint64_t out[8]);
/* wall seconds of the decoder's three host phases so far: [0] bitstream parse, [1] hand-over of the pictures' work (submit),
 * [2] waiting for / copying frames back */
void mpeghost_video_phase_seconds(void *video, double out[3]);

/* ---- Audio (audio.go:83 NewAudio, :163 Decode); format: 0 F32N, 1 F32NLR, 2 F32, 3 S16 (audio.go:12-23) */
void *mpeghost_audio_open(void *device, const uint8_t *data, size_t len, int fma_mode, int format);
void *mpeghost_audio_open_backend(void *backend, const uint8_t *data, size_t len, int format);
void mpeghost_audio_close(void *audio);
int mpeghost_audio_samplerate(void *audio);
int mpeghost_audio_channels(void *audio);
const void *mpeghost_audio_decode(void *audio, double *time);        /* 2304 samples of the format, or NULL at the end */
double mpeghost_audio_time(void *audio);                             /* audio.go:136 */
int mpeghost_audio_has_ended(void *audio);                           /* audio.go:156 */
void mpeghost_audio_rewind(void *audio);                             /* audio.go:149: the V ring stays */
void mpeghost_audio_set_lookahead(void *audio, int on);              /* Audio.Decode's one frame ahead (as the video decoder's) */

/* ---- MPEG (mpeg.go:85 New over a program stream, :356 Decode, :416 DecodeVideo, :438 DecodeAudio, :460/:524 seeks) */
void *mpeghost_mpeg_open(void *device, const uint8_t *data, size_t len);
void *mpeghost_mpeg_open_backends(void *(*make_video)(void), void *(*make_audio)(int), const uint8_t *data, size_t len);
void mpeghost_mpeg_close(void *mpeg);
void mpeghost_mpeg_info(void *mpeg, int out[6]); /* video streams, audio streams, width, height, samplerate, channels */
double mpeghost_mpeg_framerate(void *mpeg);
void mpeghost_mpeg_set_enabled(void *mpeg, int video, int audio);    /* mpeg.go:175, :250 */
void mpeghost_mpeg_get_enabled(void *mpeg, int out[2]);              /* mpeg.go:170, :245 */
void mpeghost_mpeg_set_audio_stream(void *mpeg, int stream_index);   /* mpeg.go:271: 0..3, others ignored */
void mpeghost_mpeg_set_loop(void *mpeg, int loop);                   /* mpeg.go:343 */
int mpeghost_mpeg_loop(void *mpeg);                                  /* mpeg.go:338 */
void mpeghost_mpeg_rewind(void *mpeg);                               /* mpeg.go:323 */
int mpeghost_mpeg_decode_video(void *mpeg, mpeghost_frame *out);
const float *mpeghost_mpeg_decode_audio(void *mpeg, double *time);
int mpeghost_mpeg_has_ended(void *mpeg);
int mpeghost_mpeg_take_done(void *mpeg);                             /* mpeg.go:155 Done(): 1 once after the stream ended without looping (the value on the channel) */
int mpeghost_mpeg_audio_format(void *mpeg);                          /* mpeg.go:229: 0 F32N, 1 F32NLR, 2 F32, 3 S16 */
void mpeghost_mpeg_set_audio_format(void *mpeg, int format);         /* mpeg.go:234 */
double mpeghost_mpeg_audio_lead_time(void *mpeg);                    /* mpeg.go:301, seconds */
void mpeghost_mpeg_set_audio_lead_time(void *mpeg, double seconds);  /* mpeg.go:308 */
int mpeghost_mpeg_probe(void *mpeg, size_t probe_size);
int mpeghost_mpeg_has_headers(void *mpeg);
double mpeghost_mpeg_duration(void *mpeg);
double mpeghost_mpeg_time(void *mpeg);
double mpeghost_mpeg_audio_time(void *mpeg);
double mpeghost_mpeg_video_time(void *mpeg);
void mpeghost_mpeg_count_callbacks(void *mpeg, int video, int audio); /* SetVideoCallback / SetAudioCallback that count */
void mpeghost_mpeg_callback_counts(void *mpeg, int out[2]);
void mpeghost_mpeg_decode(void *mpeg, double tick);
int mpeghost_mpeg_seek(void *mpeg, double seconds, int exact);
int mpeghost_mpeg_seek_frame(void *mpeg, double seconds, int exact, mpeghost_frame *out);

/* ---- VideoBatch: many independent video streams on ONE device, one device call per tick */
void *mpeghost_batch_open(void *device, uint32_t n_streams);
void *mpeghost_batch_open_store(void *store, uint32_t n_streams);    /* test stores; takes ownership */
void mpeghost_batch_close(void *batch);
int mpeghost_batch_add_stream(void *batch, const uint8_t *data, size_t len);
int mpeghost_batch_decode_all(void *batch, int fetch);               /* frames produced this tick, -1 error */
/* parse threads of the batch's pool.  A REQUEST: never more than the CPU time the process gets (its affinity mask capped by the
 * cgroup's CPU-time quota, rounded up — threads beyond that are throttled in turn and slow every round down); n = 0: as many as fit. */
void mpeghost_batch_set_threads(void *batch, uint32_t n);
uint32_t mpeghost_batch_threads(void *batch);                        /* what the request became */
double mpeghost_effective_cores(void);                               /* that CPU time, in cores */
/* its cgroup part: the tightest CPU-time quota (cores; 0 = none) of the process's cgroup and every ancestor (v2 cpu.max, v1 cfs
 * quota / period), found through /proc/self/cgroup; root / proc_file: stand-ins for /sys/fs/cgroup and that file (NULL = the real ones) */
double mpeghost_cgroup_quota_cores(const char *root, const char *proc_file);
int mpeghost_batch_frame(void *batch, uint32_t stream, mpeghost_frame *out);
void mpeghost_batch_counters(void *batch, uint64_t out[2]);          /* device submits, pictures queued */
void mpeghost_batch_phase_seconds(void *batch, double out[4]);       /* parse rounds, stage begin, puts, commits */
/* staged submits of sparse pictures validated + packed on the host (default: a refused picture fails the decode_all that sent it) or,
 * with on = 1 (the default since round 6), on the DEVICE (mpeghip_video_stage_begin_device): errors are then DEFERRED, per picture, to
 * the next round's start (the next mpeghost_batch_decode_all) / mpeghost_batch_sync */
void mpeghost_batch_set_device_pack(void *batch, int on);
int mpeghost_batch_sync(void *batch);                                /* wait; -1 + mpeghost_last_error(): a device-packed commit's deferred error */
/* streams whose picture the device refused, as named by the refusal last reported (decode_all / sync returning -1): a refusal is
 * per picture — the other streams' pictures of that commit were reconstructed, and it is reported BEFORE the next round parses */
uint32_t mpeghost_batch_refused_streams(void *batch, uint32_t *out, uint32_t cap);
int mpeghost_batch_device_pack(void *batch);                         /* 1 (the default): sparse pictures are validated and packed on the device */
void mpeghost_batch_debug_damage_next_picture(void *batch, uint32_t stream); /* test hook: that stream's next picture arrives damaged */
void mpeghost_batch_numa_pins(void *batch, uint32_t out[2]);         /* pool threads asked to bind to the NUMA node, bindings that failed */

/* ---- ShardedVideoBatch: streams sharded over SEVERAL devices (stream s -> device s mod G), one host thread and one
 * VideoBatch per device, no collective */
void *mpeghost_sharded_open(void *const *devices, uint32_t n_devices, uint32_t n_streams);
void *mpeghost_sharded_open_stores(void *const *stores, uint32_t n_stores, uint32_t n_streams); /* test stores; takes ownership */
void mpeghost_sharded_close(void *sharded);
int mpeghost_sharded_add_stream(void *sharded, const uint8_t *data, size_t len);
void mpeghost_sharded_set_threads(void *sharded, unsigned n);       /* parse threads of ALL shards together (0 = as many as fit): divided among them */
uint32_t mpeghost_sharded_threads(void *sharded);                   /* what the shards' pools add up to */
int mpeghost_sharded_decode_all(void *sharded, int fetch);
void mpeghost_sharded_set_device_pack(void *sharded, int on);        /* mpeghost_batch_set_device_pack of every shard */
int mpeghost_sharded_sync(void *sharded);                            /* mpeghost_batch_sync of every shard: -1 + the first deferred error */
int mpeghost_sharded_frame(void *sharded, uint32_t stream, mpeghost_frame *out);
uint32_t mpeghost_sharded_device_of(void *sharded, uint32_t stream);
void mpeghost_sharded_counters(void *sharded, uint32_t shard, uint64_t out[2]);

/* ---- AudioBatch: many MP2 streams, one synthesis call per tick */
void *mpeghost_audio_batch_open(void *device, uint32_t n_streams, int format, int fma_mode);
void *mpeghost_audio_batch_open_store(void *store, uint32_t n_streams, int format, int fma_mode);
void mpeghost_audio_batch_close(void *batch);
int mpeghost_audio_batch_add_stream(void *batch, const uint8_t *data, size_t len);
int mpeghost_audio_batch_decode_all(void *batch);
int mpeghost_audio_batch_decode_stream(void *batch, uint32_t stream); /* AudioBatch::Stream(i)->Decode(): 1 frame of one stream outside the tick; its samples arrive with the next decode_all */
const void *mpeghost_audio_batch_samples(void *batch, uint32_t stream, double *time, const void **right);
uint64_t mpeghost_audio_batch_device_calls(void *batch);
void mpeghost_audio_batch_set_threads(void *batch, uint32_t n);   /* host threads of the parse (default 1): AudioBatch::SetThreads */

/* ---- Demux (demux.go:61 NewDemux, :216 Seek) */
void *mpeghost_demux_open(const uint8_t *data, size_t len);
void mpeghost_demux_close(void *demux);
double mpeghost_demux_start_time(void *demux, int type);
double mpeghost_demux_duration(void *demux, int type);
int mpeghost_demux_probe(void *demux, size_t probe_size);
void mpeghost_demux_streams(void *demux, int out[2]);
void mpeghost_demux_rewind(void *demux);
int mpeghost_demux_decode(void *demux, double *pts, size_t *len, const uint8_t **data);
int mpeghost_demux_seek(void *demux, double seconds, int type, int force_intra, double *pts, size_t *len, const uint8_t **data);

#ifdef __cplusplus
}
#endif
#endif /* MPEGHOST_H */
