/*
 * oracle_desc.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Drives the oracle's restated reference functions (mpeg_oracle.c) from the
 * descriptor format of include/mpeghip.h, so that the HIP path and the CPU
 * restatement can be fed the very same batches.  Also the "port" CPU baseline
 * that bench.py times.
 */
#ifndef ORACLE_DESC_H
#define ORACLE_DESC_H

#include "mpeg_oracle.h"
#include "mpeghip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A frame store like mpeghip_video's: n_streams * 3 slots, zero-initialised,
 * default quantiser matrices. */
typedef struct orc_store orc_store;
orc_store *orc_store_open(int width, int height, uint32_t n_streams);
void       orc_store_close(orc_store *s);
orc_frame *orc_store_frame(orc_store *s, uint32_t stream, uint32_t slot);
void       orc_store_set_quant(orc_store *s, uint32_t stream, const uint8_t intra[64], const uint8_t non_intra[64]);

/* Reconstruct a submit in descriptor order with the reference's routines:
 * orc_copy_macroblock, the dequantisation of video.go:719-744, orc_idct with
 * the variant the reference would pick (n = last non-zero in scan order + 1;
 * DC-only fast path for n == 1) and the *ToDest writers.
 * n_threads > 1 splits the PICTURES over that many threads (pictures of one
 * submit are independent).  Returns 0, or -1 on a range violation. */
int orc_store_submit(orc_store *s, const mpeghip_pic_desc *pics, uint32_t n_pics,
                     const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                     const void *coefs, size_t coef_bytes, int n_threads);

/* Audio synthesis state of one stream + the reference loop audio.go:378-422. */
typedef struct orc_synth {
    float v[2][1024];
    int32_t vpos;
} orc_synth;
/* samples: int32 [n_frames][2][36][32]; out: [n_frames][2304] of the format's type. */
void orc_synth_frames(orc_synth *st, const int32_t *samples, uint32_t n_frames,
                      int format, int fma, void *out);

#ifdef __cplusplus
}
#endif
#endif
