/*
 * mpeghip.h — C ABI of libmpeghip, the MI355X (gfx950) reconstruction core for
 * MPEG-1 video / MPEG-1 Audio Layer II decode.
 *
 * This is the drop-in boundary for the hot path of gen2brain/mpeg (reference,
 * file:line relative to its repository root).  The reference has no FFI seam of
 * its own: its hot functions are plain package functions called once per
 * macroblock / per 32-sample sub-block, which is far too fine to cross cgo.
 * The boundary is therefore cut one level up: the (serial, CPU) bitstream
 * parser records work into flat descriptor arrays and hands a whole picture
 * (or a batch of pictures from many independent streams) / a whole audio frame
 * (or many) to the device in one call.  Each entry point below names the
 * reference code it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; all multi-byte fields little-endian.
 *   - every function returns MPEGHIP_OK (0) or a negative MPEGHIP_ERR_* code;
 *     mpeghip_last_error() gives a human readable message for the calling thread.
 *   - nothing throws, nothing aborts.  There is NO CPU fallback: without a
 *     usable HIP device mpeghip_ctx_create fails with MPEGHIP_ERR_NO_DEVICE.
 *   - a context is bound to one GPU and one HIP stream; contexts are
 *     independent (one per GPU / per host thread), not thread-safe.
 *   - host buffers passed in are only read/written during the call (cgo rule:
 *     C never retains Go memory).  Use mpeghip_pinned_alloc for staging that
 *     should be DMA-able.
 */
#ifndef MPEGHIP_H
#define MPEGHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the sparse hand-over's snapshot blocks carry a count word, a sparse picture's macroblocks name their words in order;
 *    the device-packed stage (mpeghip_video_stage_begin_device) and its deferred errors; chroma as Cb|Cr pairs in device memory */
/* 3: (round 6) asynchronous read-back and synthesis for a lone decoder that works one picture / frame ahead —
 *    mpeghip_video_read_planes_async / _read_wait, mpeghip_audio_synth_async / _synth_wait / _undo_last; a device-packed commit's
 *    refusal is per picture (the commit's other pictures are reconstructed: mpeghip_video_verdict / _refused); and
 *    mpeghip_ctx_pci_bus_id (added during round 5 without a bump); mpeghip_video_host_mirror / _mirror_async (a lone decoder's frames
 *    written linearly into pinned host memory by the reconstruction launch itself).  Nothing else of version 2 changed. */
#define MPEGHIP_ABI_VERSION 3

#define MPEGHIP_OK             0
#define MPEGHIP_ERR_INVALID   (-1) /* bad argument / malformed descriptor            */
#define MPEGHIP_ERR_NO_DEVICE (-2) /* no HIP device / wrong architecture              */
#define MPEGHIP_ERR_HIP       (-3) /* HIP runtime error (message has the details)     */
#define MPEGHIP_ERR_OOM       (-4) /* host or device allocation failed                */
#define MPEGHIP_ERR_RANGE     (-5) /* a motion vector reads outside the frame buffer  */

typedef struct mpeghip_ctx   mpeghip_ctx;
typedef struct mpeghip_video mpeghip_video;
typedef struct mpeghip_batch mpeghip_batch;
typedef struct mpeghip_audio mpeghip_audio;

/* ------------------------------------------------------------------ context */

/* device: HIP device ordinal.  stream: an existing hipStream_t (e.g. the
 * caller's torch stream) or NULL to let the context create its own. */
int  mpeghip_ctx_create(int device, void *stream, mpeghip_ctx **out);
void mpeghip_ctx_destroy(mpeghip_ctx *ctx);
/* The host NUMA node the context's GPU is attached to (sysfs: its PCI function's numa_node), -1 if the platform does not
 * say.  The multi-GPU driver pins each device's host threads (bitstream parse, staged puts) to that node's cores. */
int  mpeghip_ctx_numa_node(const mpeghip_ctx *ctx);
/* The PCI address of the context's GPU ("0000:c1:00.0", lower case) into out[0..cap): what tells two ranks on one physical
 * device apart from two ranks on two (HIP ordinals do not: HIP_VISIBLE_DEVICES renumbers them per process).  bench.py's N > 1
 * line lists every rank's and counts the distinct ones as n_gpus.  MPEGHIP_ERR_INVALID if cap is too small. */
int  mpeghip_ctx_pci_bus_id(const mpeghip_ctx *ctx, char *out, size_t cap);
int  mpeghip_ctx_sync(mpeghip_ctx *ctx);               /* wait for all queued work   */
int  mpeghip_device_count(void);                       /* <0 on error                */
const char *mpeghip_last_error(void);                  /* thread-local, never NULL   */
int  mpeghip_abi_version(void);

/* Pinned (page-locked) host staging memory, hipHostMalloc-backed. */
void *mpeghip_pinned_alloc(mpeghip_ctx *ctx, size_t bytes);
void  mpeghip_pinned_free(mpeghip_ctx *ctx, void *p);

/* Stream-ordered timing helpers (hipEvent on the context's stream). */
int mpeghip_timer_start(mpeghip_ctx *ctx);
int mpeghip_timer_stop_ms(mpeghip_ctx *ctx, float *ms); /* records, syncs, returns elapsed */

/* -------------------------------------------------------------------- video */

/*
 * Frame store (replaces Video.initFrame, video.go:333-372, and the three
 * rotating Frame values frameCurrent/frameForward/frameBackward,
 * video.go:97-99).  For every stream the handle owns 3 frame "slots",
 * zero-initialised.  SEEN THROUGH THIS ABI (read_planes / write_planes /
 * hash_slots / read_rgba, and every prediction, including the reference's
 * reads past the end of a plane row or plane, video_noasm.go:48-80) a slot is
 * the reference's `base` slice:
 *
 *     Y  [luma_w  * luma_h ]   luma_w  = mb_w*16, luma_h  = mb_h*16
 *     Cb [chroma_w*chroma_h]   chroma_w= mb_w*8,  chroma_h= mb_h*8
 *     Cr [chroma_w*chroma_h]
 *     pad[luma_w * 16]         zero; half-pel reads past a plane end land here
 *
 * IN DEVICE MEMORY the planes are stored tiled: luma as
 * 16x16 tiles of 256 bytes in macroblock raster order from offset 0; behind
 * them, from luma_bytes on, one 128-byte PAIR per macroblock, its 8x8 Cb block
 * (64 bytes) followed by its 8x8 Cr block.  Byte (x, y) of the luma plane lives at
 * ((y/16)*mb_w + x/16)*256 + (y%16)*16 + x%16; byte (x, y) of Cb at
 * luma_bytes + ((y/8)*mb_w + x/8)*128 + (y%8)*8 + x%8, of Cr 64 bytes further
 * on.  Only mpeghip_video_slot_devptr exposes that; the RGBA image is linear.
 *
 * Which slot plays current/forward/backward is the caller's business (the
 * parser mirrors the rotation of video.go:406-409/430-433) and is stated per
 * picture in mpeghip_pic_desc.
 */
typedef struct mpeghip_video_info {
    uint32_t width, height;          /* display size                              */
    uint32_t mb_w, mb_h;             /* macroblocks per row / column              */
    uint32_t luma_w, luma_h;         /* padded plane sizes                        */
    uint32_t chroma_w, chroma_h;
    uint32_t n_streams;
    uint32_t reserved;
    uint64_t luma_bytes, chroma_bytes;
    uint64_t frame_bytes;            /* Y+Cb+Cr+pad, the reference's len(base)    */
    uint64_t frame_stride;           /* device bytes between consecutive slots    */
    uint64_t rgba_bytes;             /* width*height*4                            */
} mpeghip_video_info;

#define MPEGHIP_SLOTS 3

int  mpeghip_video_open(mpeghip_ctx *ctx, uint32_t width, uint32_t height,
                        uint32_t n_streams, mpeghip_video **out);
void mpeghip_video_close(mpeghip_video *v);
int  mpeghip_video_info_get(const mpeghip_video *v, mpeghip_video_info *info);

/* Quantiser matrices of one stream, row-major natural order (already
 * de-zigzagged, as video.go:291-312 stores them).  Defaults are the MPEG-1
 * default matrices (video.go:1055-1075). */
int mpeghip_video_set_quant(mpeghip_video *v, uint32_t stream,
                            const uint8_t intra[64], const uint8_t non_intra[64]);

/* Tuning knob: which instance of the reconstruction kernel a submit / batch of this handle runs on.  The library picks
 * per batch (AUTO): batches with more than an eighth of their coded blocks dense (more than 32 non-zero
 * levels) take the instance that is built for vector-ALU-bound work (the IDCT's transposition through LDS, a short dequantisation
 * of dense units: MPEGHIP_TILE_INT32 — the name is history, both instances keep an int16 coefficient tile and run 8 waves per
 * SIMD); the others the one that transposes across lanes (MPEGHIP_TILE_INT16); and a launch small enough to leave most of the device
 * empty (one or two 1080p pictures) runs on a third kernel that puts four waves on every chunk of 4 macroblocks (recon_wide_kernel: the
 * launch lasts one chunk's chain of dependent steps, which four waves walk in parallel).  Results are identical bit for bit
 * either way; the override pins one of the two instances at any size and exists for measurements (bench.py --tile) and for
 * callers who know their streams.  No reference counterpart. */
#define MPEGHIP_TILE_AUTO  0
#define MPEGHIP_TILE_INT16 1
#define MPEGHIP_TILE_INT32 2
int mpeghip_video_set_tile_policy(mpeghip_video *v, int policy);

/* One picture of one stream. */
typedef struct mpeghip_pic_desc {
    uint32_t stream;      /* stream index inside the video handle                    */
    uint8_t  cur;         /* slot written by this picture (frameCurrent)             */
    uint8_t  fwd;         /* slot read for forward prediction (frameForward)         */
    uint8_t  bwd;         /* slot read for backward prediction (frameBackward)       */
    uint8_t  flags;       /* MPEGHIP_PIC_*; undefined bits are refused               */
    uint32_t mb_first;    /* first macroblock descriptor of this picture             */
    uint32_t mb_count;    /* number of macroblock descriptors                        */
} mpeghip_pic_desc;       /* 16 bytes */

#define MPEGHIP_PIC_RGBA 0x01u /* also colour-convert every written macroblock into
                                  the cur slot's RGBA image (fused Frame.RGBA())   */
/* MPEGHIP_PIC_SPARSE (0x02, below): the picture's coefficient data is in the sparse form */

/*
 * One macroblock.  Replaces one trip through decodeMacroblock's reconstruction
 * half: predictMacroblock (video.go:608-637) -> copyMacroblock
 * (video_noasm.go:28-43 / video_amd64.s / video_arm64.s), then per coded block
 * the dequantise+premultiply tail of decodeBlock (video.go:719-744), idct
 * (video.go:801-928) and copy/add*ToDest (video.go:943-1002).
 *
 * Motion: mv_x/mv_y are the luma vector in half-pel units AFTER the full-pel
 * doubling of video.go:612-624.  The reference never averages two
 * predictions: for a B macroblock with both vectors it copies the forward
 * prediction and then overwrites it with the backward one (video.go:626-630),
 * so a descriptor names exactly ONE reference (the one whose bytes survive).
 * Chroma vectors are derived on the device as mv/2 truncated toward zero
 * (video_noasm.go:35-36).
 *
 * Coefficients: the macroblock's coded blocks (bit 5-b of cbp set <=> block b,
 * b = 0..3 luma in raster order, 4 = Cb, 5 = Cr) follow each other in block
 * order starting at coefs + coef_off*128 bytes.  Each block is 64 values in
 * COLUMN-major order (value for row r, column c at index c*8+r):
 *   - default: int16 quantised levels as read from the bitstream (0 = not
 *     present), 128 bytes; for intra blocks index 0 holds the reconstructed DC
 *     (predictor+differential, video.go:653-669) instead.  The device performs
 *     the reference's dequantisation (video.go:719-741) with qscale and the
 *     stream's matrices, the premultiply (video.go:744) and `<<8` for intra DC
 *     (video.go:672).
 *   - MPEGHIP_MB_COEF_RAW: int32 already dequantised+premultiplied values, 256
 *     bytes (2 units).  This is a verbatim snapshot of the reference's live
 *     `blockData`, needed where blockData carries stale coefficients from an
 *     earlier invalid block (video.go:711-714 returns before the clears) or a
 *     value that the int16 form cannot express.  The emitter applies the
 *     sparse-IDCT masking (rows<4 & cols<4 when n<10, video.go:807-866) and the
 *     DC-only rule (n==1, video.go:774-777/787-790) before snapshotting, so the
 *     device runs ONE full IDCT for every coded block.
 * Intra macroblocks write only their coded blocks (an invalid block leaves the
 * old pixels, video.go:711-714); inter macroblocks write all 384 bytes.
 */
typedef struct mpeghip_mb_desc {
    uint32_t pic;         /* index into the submit's mpeghip_pic_desc array          */
    uint16_t mb_x, mb_y;  /* macroblock column / row                                 */
    int16_t  mv_x, mv_y;  /* half-pel luma motion vector                             */
    uint8_t  flags;       /* MPEGHIP_MB_*                                            */
    uint8_t  cbp;         /* coded (and valid) block pattern, bit 5-b <=> block b    */
    uint8_t  qscale;      /* quantiser_scale 1..31 (ignored for COEF_RAW)            */
    uint8_t  reserved0;
    uint32_t coef_off;    /* first coefficient block, in 128-byte units              */
    uint32_t reserved[3]; /* 0                                                       */
} mpeghip_mb_desc;        /* 32 bytes */

#define MPEGHIP_MB_INTRA    0x01u /* no prediction, coded blocks overwrite           */
#define MPEGHIP_MB_REF_FWD  0x02u /* predict from pic.fwd                            */
#define MPEGHIP_MB_REF_BWD  0x04u /* predict from pic.bwd                            */
#define MPEGHIP_MB_COEF_RAW 0x08u /* int32 premultiplied coefficient blocks          */

#define MPEGHIP_COEF_UNIT 128u

/* Validate the descriptors, pack them into the library's device format and
 * reconstruct, stream ordered.
 * Asynchronous: the packed form is written into one of two pinned staging buffers and the call returns
 * with the H2D copy and the kernel in flight, so the caller parses picture N+1 while picture N is
 * reconstructed (the arrays may be reused at once; a third submit waits for the first).
 * Macroblocks of one submit run concurrently, so a submit is refused with
 * MPEGHIP_ERR_INVALID (nothing is launched) when
 *   - one picture addresses a macroblock position twice (the emitter starts a
 *     new submit when a damaged stream does, so "last writer in bitstream
 *     order" is kept by stream order),
 *   - a picture predicts from the slot it writes (cur == the named reference),
 *   - two pictures of the same stream write the same slot, or one reads a
 *     slot another one of the same submit writes,
 *   - its coded blocks name more coefficient units (counted with repetition)
 *     than `coefs` holds: blocks may share units only as far as there are as
 *     many units as blocks name (the packed form's buffers are sized from
 *     coef_bytes).
 * Returns MPEGHIP_ERR_RANGE (nothing is launched) if any prediction would
 * read outside [plane start, end of pad) — the reference panics there. */
int mpeghip_video_submit(mpeghip_video *v,
                         const mpeghip_pic_desc *pics, uint32_t n_pics,
                         const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                         const void *coefs, size_t coef_bytes);

/* The same submit, staged picture by picture from several host threads (the many-stream emitter:
 * one parser thread per stream, mpeg_amd/host/batch.cpp).  `begin` reserves room for n_pics pictures of
 * the given sizes in the next pinned staging buffer; `put` validates picture i and writes it there —
 * thread-safe for distinct i, no device access, mbs[].pic is ignored (all belong to picture i), coef_off is
 * relative to the picture's own `coefs`; `commit` (the thread that owns the context) sends the buffer and
 * reconstructs, asynchronous like mpeghip_video_submit, and ends the stage.  A failed `put` makes `commit`
 * fail with that error and launch nothing.  One stage per video at a time; no other submit in between.
 * The rules of mpeghip_video_submit about overlapping macroblocks and dependent pictures apply. */
typedef struct mpeghip_stage mpeghip_stage;
int mpeghip_video_stage_begin(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *coef_bytes,
                              mpeghip_stage **out);
int mpeghip_video_stage_put(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic,
                            const mpeghip_mb_desc *mbs, const void *coefs);
int mpeghip_video_stage_commit(mpeghip_stage *s);

/* The SPARSE hand-over: a picture's coefficient data as the reference's VLC loop produces it — one (position, level)
 * pair per coded coefficient (video.go:680-745: `n += run`, de-zigzag, level) — instead of 128-byte units of mostly
 * zeros.  The device format is sparse too (one dword per non-zero level), so the parser's pairs become device entries
 * with one OR each; nothing is densified and nothing is searched for zeros again.
 * A picture is in this form when its descriptor carries MPEGHIP_PIC_SPARSE; every entry point that takes pictures
 * (submit, stage_put, batch_upload[_replicated]) takes either form, picture by picture.  For a sparse picture
 *   mbs[k].coef_off   counts DWORDS from the start of `coefs` (not 128-byte units), and from there on lie the data of
 *                     macroblock k's coded blocks in block order (cbp bit 5 first):
 *     a block:           a count word n (0..64), then n pair words MPEGHIP_PAIR(level, position);
 *                        position = column * 8 + row (the order of a coefficient unit), each position at
 *                        most once per block (not checked — a parser cannot produce it; the result of such a
 *                        block is unspecified: one of the two levels wins); level: any int16.  An INTRA block's first pair is its DC
 *                        (position 0, always present; `<< 8` in the reference, video.go:672).  A level of 0
 *                        is a CODED zero: the reference dequantises it to +-1 (video.go:719-736) — the form
 *                        that units (0 = absent) need a snapshot for.
 *     a snapshot block   (macroblock flag MPEGHIP_MB_COEF_RAW): the count word 64, then 64 int32 values, column-major
 *                        (ABI version 2: every block begins with a count word)
 *   macroblocks name their words IN ORDER and without overlap: mbs[k].coef_off is at least the end of macroblock k-1's
 *                     data and at most the picture's dwords — for every macroblock, coded blocks or not.  (So the packed form of
 *                     a picture is never longer than its input, and the device-side packer places a chunk's words by the
 *                     offset of its first macroblock.)
 *   coef_bytes        a multiple of 4 (of 128 as soon as one picture of the call is in the unit form)
 * Malformed block data (a count beyond 64 — or other than 64 for a snapshot block —, a block that ends behind the buffer,
 * macroblocks out of order or overlapping, bits outside the two fields of a pair, an intra block without its DC first) is
 * refused with MPEGHIP_ERR_INVALID, nothing launched; so are pictures of one submit that name the same words beyond the
 * room the packed form was given (the buffer's dwords).
 * The three functions below are the same calls with the flag set for the caller and sizes in dwords:
 *   mpeghip_video_stage_begin_sparse / _put_sparse   the many-stream emitter (one parser thread per stream)
 *   mpeghip_video_submit_sparse                      ONE picture per call: the single-stream decoder's flush */
#define MPEGHIP_PIC_SPARSE 0x02u
#define MPEGHIP_PAIR(level, position) (((uint32_t)(uint16_t)(level) << 16) | ((uint32_t)(position) << 2))
int mpeghip_video_stage_begin_sparse(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *n_words,
                                     mpeghip_stage **out);
int mpeghip_video_stage_put_sparse(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic,
                                   const mpeghip_mb_desc *mbs, const uint32_t *words);
int mpeghip_video_submit_sparse(mpeghip_video *v, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                                const uint32_t *words, size_t n_words);

/* The DEVICE-PACKED stage: the same staged submit with the library's host work reduced to a copy.  The pictures' arrays —
 * mpeghip_pic_desc, mpeghip_mb_desc, the sparse form's count / pair words — travel to the device as they are, and a kernel in
 * front of the reconstruction validates them and packs them into the device format there: what
 * validate + pack cost a host core per picture (0.27 ms for a typical 1080p picture) is what kept the staged hand-over at
 * 44 % of the PCIe link's rate.  Sparse pictures only (mpeghip_video_stage_put on such a stage fails).
 *   mpeghip_video_stage_begin_device   as _begin_sparse: n_words[i] dwords for picture i
 *   mpeghip_video_stage_put_sparse     copies picture i's arrays into the pinned staging buffer (thread-safe for distinct i)
 *   mpeghip_video_stage_map            OR: where picture i's arrays live in that buffer — n_mbs[i] descriptors, n_words[i]
 *                                      dwords — so that a parser writes them there in the first place (pinned host memory of
 *                                      the library, valid until the commit), then
 *   mpeghip_video_stage_put_mapped     hands over the picture's descriptor and marks picture i complete; no copy at all
 *   mpeghip_video_stage_commit         one H2D copy, pack, reconstruct; returns with all of it in flight
 * DEFERRED ERRORS.  put / put_mapped check the picture descriptor only.  Everything the host path refuses with
 * MPEGHIP_ERR_INVALID / MPEGHIP_ERR_RANGE at `put` or `commit` — macroblock fields, vectors that leave the frame buffer, a
 * position addressed twice, malformed block data, a picture that reads what another picture of its stream in the same commit
 * writes — is found by the device AFTER the commit has returned MPEGHIP_OK.  The unit of failure is the PICTURE, as in the
 * reference (video.go:374-460; ABI 3 — until then a refusal dropped the whole commit): a refused picture is not reconstructed
 * (its stream's frame store is as before it), every OTHER picture of the commit — they belong to other streams — is; later
 * commits run as queued.  The error — the first refused picture, its stream and macroblock, and how many pictures were refused,
 * in mpeghip_last_error() — is returned ONCE by the next call on the handle that waits for the device: mpeghip_video_verdict,
 * mpeghip_video_sync, read_planes, read_rgba, hash_slots, or the stage_begin / submit that reuses the commit's staging buffer (the
 * second one after it); mpeghip_video_refused then names the refused pictures and their streams.  A caller that must know
 * before it goes on calls mpeghip_video_verdict: it waits for the validation only (the packing kernels in front of the
 * reconstruction), not for the reconstruction.
 * The sparse form's rule that macroblocks name their words in order and without overlap is what lets the device place a
 * chunk's words without a scan; cbp == 0 macroblocks take part in it (their coef_off: anything from the previous
 * macroblock's end to the picture's n_words). */
int mpeghip_video_stage_begin_device(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *n_words,
                                     mpeghip_stage **out);
int mpeghip_video_stage_map(mpeghip_stage *s, uint32_t i, mpeghip_mb_desc **mbs, uint32_t **words);
int mpeghip_video_stage_put_mapped(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic);
/* Wait for everything queued on this handle; returns (once) the deferred error of a device-packed commit, if any. */
int mpeghip_video_sync(mpeghip_video *v);
/* (ABI 3) Wait until the device has VALIDATED the device-packed commits queued so far — not until it has reconstructed them —
 * and return (once) their deferred error, if any.  What a batch of decoders calls before it parses its next pictures. */
int mpeghip_video_verdict(mpeghip_video *v);
/* (ABI 3) The pictures refused by the error last returned for a device-packed commit: up to `cap` of them, as (index of the picture
 * in its commit, its stream), in picture order (either array may be NULL); returns how many were refused (the library names the
 * first 1 020 of a commit). */
uint64_t mpeghip_video_refused(const mpeghip_video *v, uint32_t *pics, uint32_t *streams, uint32_t cap);

/* Device-resident batches: validate + upload once, replay many times
 * (synthetic benchmark batches; a real decoder double-buffers two). */
int  mpeghip_video_batch_upload(mpeghip_video *v,
                                const mpeghip_pic_desc *pics, uint32_t n_pics,
                                const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                                const void *coefs, size_t coef_bytes,
                                mpeghip_batch **out);
/* Same descriptors replicated for `n_streams` streams on the device: the batch
 * describes stream 0 only (all pics must have stream==0); stream s gets a copy
 * whose pictures address stream s.  Costs one descriptor set in HBM per stream,
 * exactly as independent streams would. */
int  mpeghip_video_batch_upload_replicated(mpeghip_video *v,
                                const mpeghip_pic_desc *pics, uint32_t n_pics,
                                const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                                const void *coefs, size_t coef_bytes,
                                uint32_t n_streams, mpeghip_batch **out);
int  mpeghip_video_batch_run(mpeghip_video *v, const mpeghip_batch *b);
void mpeghip_video_batch_free(mpeghip_batch *b);
/* Algorithmic HBM bytes of one run of the batch:
 * sum over macroblocks of 32 + coefficient bytes + reference window + 384
 * (+1024 per macroblock of pictures flagged MPEGHIP_PIC_RGBA). */
uint64_t mpeghip_video_batch_alg_bytes(const mpeghip_batch *b);
uint64_t mpeghip_video_batch_mbs(const mpeghip_batch *b);
/* Bytes of the batch in the library's device format (picture descriptors + chunks + words) as uploaded, before any
 * replication: what a submit of the same pictures moves over PCIe. */
uint64_t mpeghip_video_batch_device_bytes(const mpeghip_batch *b);

/* Plane access (replaces reading Frame.Y/Cb/Cr.Data, video.go:17-19).  Sizes
 * are luma_bytes / chroma_bytes.  Synchronous. `pad` (luma_w*16 bytes) may be
 * NULL. */
int mpeghip_video_read_planes(mpeghip_video *v, uint32_t stream, uint32_t slot,
                              uint8_t *y, uint8_t *cb, uint8_t *cr);
/* The same read-back, ASYNCHRONOUS (ABI 3; replaces nothing in the reference — it is what lets the Go shim's Video.Decode,
 * video.go:209-268, parse picture N+1 while picture N is on the device): queued behind everything submitted so far, the
 * planes of (stream, slot) as they are at that point land in `dst` — luma | Cb | Cr in the reference's linear layout,
 * luma_bytes + 2 * chroma_bytes — and the call returns at once.  `dst` should be memory of mpeghip_pinned_alloc (the untiling
 * kernel then stores straight into it; other memory works through a copy).  *ticket names the read-back;
 * mpeghip_video_read_wait(ticket) blocks until it is in `dst` (and reports a deferred device-packed verdict, like every
 * synchronising call).  Read-backs complete in the order they were queued. */
int mpeghip_video_read_planes_async(mpeghip_video *v, uint32_t stream, uint32_t slot, uint8_t *dst, uint64_t *ticket);
int mpeghip_video_read_wait(mpeghip_video *v, uint64_t ticket);
/* The HOST MIRROR (round 6, within ABI 3; for a lone decoder's store — what makes the Go shim's Video.Decode, video.go:209-268,
 * one launch per picture): mpeghip_video_host_mirror(v, 1) gives every (stream, slot) a copy of its planes in pinned host memory,
 * in the reference's linear layout (luma | Cb | Cr, luma_bytes + 2 * chroma_bytes; at most 1 GiB in all), and from then on the
 * reconstruction launch itself writes every macroblock there as well — for submits small enough for the library's four-waves-per-
 * chunk kernel (up to two 1080p pictures; no MPEGHIP_PIC_RGBA picture among them).  mpeghip_video_mirror_async names the copy of
 * (stream, slot) — *planes, valid memory until the mirror is switched off or the store closed — and a ticket:
 * mpeghip_video_read_wait(ticket) returns when the copy holds the slot as it is after everything submitted so far.  No untiling
 * launch and no copy are queued, unless something else wrote the slot since (a large or colour-converting submit, write_planes,
 * broadcast_slot, or nothing yet): then one untiling launch into the copy makes it right again.  The copy of a slot is overwritten by
 * the next picture reconstructed into that slot — which is the lifetime the reference gives a returned *Frame (mpeg.go:413-415). */
int mpeghip_video_host_mirror(mpeghip_video *v, int on);
int mpeghip_video_mirror_async(mpeghip_video *v, uint32_t stream, uint32_t slot, const uint8_t **planes, uint64_t *ticket);
/* out[0]: mpeghip_video_mirror_async calls so far; out[1]: those that had to queue an untiling launch first (a lone decoder in its
 * stride: three, one per slot) */
void mpeghip_video_mirror_counters(const mpeghip_video *v, uint64_t out[2]);
int mpeghip_video_write_planes(mpeghip_video *v, uint32_t stream, uint32_t slot,
                               const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                               const uint8_t *pad);
/* Copy one slot of stream `src` to the same slot of streams [dst0, dst0+n). */
int mpeghip_video_broadcast_slot(mpeghip_video *v, uint32_t src, uint32_t slot,
                                 uint32_t dst0, uint32_t n);
/* FNV-1a-64 over Y||Cb||Cr (the byte order TestVideoGolden hashes,
 * mpeg_test.go:221-223) of every stream's slot, computed on the device; used
 * by full-size property tests.  out[n_streams]. */
int mpeghip_video_hash_slots(mpeghip_video *v, uint32_t slot, uint64_t *out);

/* Frame.RGBA() (video.go:31-36 -> Go image/draw YCbCr 4:2:0 -> RGBA, JFIF
 * full range, alpha 255).  Converts the slot's planes into the slot's RGBA
 * image on the device (all streams in [stream0, stream0+n)), stream ordered. */
int mpeghip_video_rgba_convert(mpeghip_video *v, uint32_t slot,
                               uint32_t stream0, uint32_t n);
/* Read the slot's RGBA image (width*height*4 bytes, stride 4*width). Synchronous. */
int mpeghip_video_read_rgba(mpeghip_video *v, uint32_t stream, uint32_t slot, uint8_t *dst);

/* Raw device addresses (for zero-copy consumers and on-device checks).  The slot is in the TILED
 * device layout described at the top of this section; the RGBA image is linear, stride 4*width. */
void *mpeghip_video_slot_devptr(mpeghip_video *v, uint32_t stream, uint32_t slot);
void *mpeghip_video_rgba_devptr(mpeghip_video *v, uint32_t stream, uint32_t slot);

/* -------------------------------------------------------------------- audio */

/*
 * MP2 synthesis (replaces the synthesis loop of Audio.decodeFrame,
 * audio.go:378-422: idct36 audio.go:492-772 — the 32-point matrixing DCT —,
 * synthWindow audio_noasm.go:8-38 / audio_amd64.s / audio_arm64.s, and the
 * output scaling audio.go:386-418).
 *
 * Input per stream and frame: the requantised sub-band samples
 * (Audio.sample after readSamples, audio.go:440-490) for both channels, laid
 * out int32 [2 ch][36 sub-blocks][32 sub-bands]; sub-block t = (part*4 +
 * granule)*3 + p in the reference's loop order.  The reference synthesises
 * both channels even for mono (audio.go:382), so does the device.
 * Per-stream state carried between calls: the V ring v[2][1024] and vPos
 * (audio.go:63,78), zero at open; Audio.Rewind does NOT clear it (audio.go:149).
 */
#define MPEGHIP_AUDIO_F32N   0 /* float32 normalised, interleaved L R (Samples.Interleaved) */
#define MPEGHIP_AUDIO_F32NLR 1 /* float32 normalised, planar: 1152 L then 1152 R            */
#define MPEGHIP_AUDIO_F32    2 /* float32 scaled by 2^31, interleaved (Samples.F32)         */
#define MPEGHIP_AUDIO_S16    3 /* int16 interleaved (Samples.S16)                           */

#define MPEGHIP_AUDIO_FMA_NONE   0 /* mul and add rounded separately: amd64 pure-Go/SSE2,
                                      golden hash 0xf1b76cdf8e6cdea5 (mpeg_test.go:194)   */
#define MPEGHIP_AUDIO_FMA_WINDOW 1 /* fused multiply-add in the window only: amd64 AVX2,
                                      golden hash 0x50f3ab75f5fb0fb5 (mpeg_test.go:195)   */

#define MPEGHIP_AUDIO_FRAME_SAMPLES 1152
#define MPEGHIP_AUDIO_FRAME_INTS    (2 * 36 * 32)

int  mpeghip_audio_open(mpeghip_ctx *ctx, uint32_t n_streams, int fma_mode,
                        mpeghip_audio **out);
void mpeghip_audio_close(mpeghip_audio *a);

/* samples: host int32 [n_streams][n_frames][2][36][32]; out: host buffer,
 * [n_streams][n_frames][2304] elements of the format's type.  Synchronous. */
int mpeghip_audio_synth(mpeghip_audio *a, const int32_t *samples, uint32_t n_frames,
                        int format, void *out);
/* Same for a subset of the streams: active[n_streams], streams with 0 sit the call out — their V ring
 * and vPos are carried over unchanged, their slots in `samples` / `out` are ignored / left alone.  What a
 * batch of decoders needs when some streams have no frame this tick (NULL = all active). */
int mpeghip_audio_synth_masked(mpeghip_audio *a, const int32_t *samples, uint32_t n_frames,
                               int format, void *out, const uint8_t *active);
/* Same with device-resident input/output (stream ordered, asynchronous). */
int mpeghip_audio_synth_device(mpeghip_audio *a, const int32_t *d_samples,
                               uint32_t n_frames, int format, void *d_out);
/* Device scratch owned by the handle, big enough for n_frames: returns device
 * pointers (benchmarks fill d_samples once and replay). */
int mpeghip_audio_device_buffers(mpeghip_audio *a, uint32_t n_frames, int format,
                                 int32_t **d_samples, void **d_out);
int mpeghip_audio_upload(mpeghip_audio *a, int32_t *d_dst, const int32_t *src, size_t n_ints);
int mpeghip_audio_download(mpeghip_audio *a, void *dst, const void *d_src, size_t bytes);
/* mpeghip_audio_synth, ASYNCHRONOUS (ABI 3; what lets the Go shim's Audio.Decode, audio.go:163-182, parse frame N+1 while
 * frame N is synthesised): `samples` and `out` laid out as for mpeghip_audio_synth, in memory of mpeghip_pinned_alloc that
 * stays untouched until mpeghip_audio_synth_wait(ticket) returns (the kernel reads and writes pinned memory in place; other
 * memory works through copies).  Launches run in the order they were queued. */
int mpeghip_audio_synth_async(mpeghip_audio *a, const int32_t *samples, uint32_t n_frames, int format, void *out, uint64_t *ticket);
int mpeghip_audio_synth_wait(mpeghip_audio *a, uint64_t ticket);
/* Forget the LAST launch: every stream's V ring and vPos are again what they were before it (one level).  For a decoder that
 * synthesised one frame ahead and is rewound: the reference's ring (never cleared by Rewind, audio.go:149-154) is what the
 * frames it RETURNED left behind. */
int mpeghip_audio_undo_last(mpeghip_audio *a);
/* V ring + vPos of one stream: v[2][1024] float32, vpos in [0,1024) multiple of 64.
 * set_state takes what get_state returned (or zeros): every 64-entry slot of Audio.v is idct36's
 * signed mirror of 32 DCT outputs (audio.go:708-771: d[48-k] == d[48+k], d[k-16] == -d[48-k],
 * d[16] == 0) and the kernel keeps only those 32; a ring that breaks the mirror is refused with
 * MPEGHIP_ERR_INVALID rather than synthesised differently from the reference. */
int mpeghip_audio_get_state(mpeghip_audio *a, uint32_t stream, float *v, int32_t *vpos);
int mpeghip_audio_set_state(mpeghip_audio *a, uint32_t stream, const float *v, int32_t vpos);

#ifdef __cplusplus
}
#endif
#endif /* MPEGHIP_H */
