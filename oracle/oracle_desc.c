/*
 * oracle_desc.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle_desc.h).
 */
#include "oracle_desc.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t kd_zigzag[64] = { /* video.go:1044-1053 */
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static const uint8_t kd_intra_q[64] = { /* video.go:1055-1064 */
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};
static const uint8_t kd_premult[64] = { /* video.go:1077-1086 */
    32, 44, 42, 38, 32, 25, 17, 9,  44, 62, 58, 52, 44, 35, 24, 12,
    42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
    32, 44, 42, 38, 32, 25, 17, 9,  25, 35, 33, 30, 25, 20, 14, 7,
    17, 24, 23, 20, 17, 14, 9,  5,  9,  12, 12, 10, 9,  7,  5,  2};

struct orc_store {
    int width, height;
    uint32_t n_streams;
    orc_frame *frames;        /* [n_streams*3] */
    uint8_t (*qm)[2][64];     /* [n_streams][intra,non-intra][64] natural order */
};

orc_store *orc_store_open(int width, int height, uint32_t n_streams)
{
    orc_store *s = (orc_store *)calloc(1, sizeof(*s));
    s->width = width;
    s->height = height;
    s->n_streams = n_streams;
    s->frames = (orc_frame *)calloc((size_t)n_streams * 3, sizeof(orc_frame));
    s->qm = (uint8_t(*)[2][64])calloc(n_streams, sizeof(*s->qm));
    for (uint32_t i = 0; i < n_streams * 3; i++)
        orc_frame_alloc(&s->frames[i], width, height);
    for (uint32_t i = 0; i < n_streams; i++) {
        memcpy(s->qm[i][0], kd_intra_q, 64);
        memset(s->qm[i][1], 16, 64);
    }
    return s;
}

void orc_store_close(orc_store *s)
{
    if (!s)
        return;
    for (uint32_t i = 0; i < s->n_streams * 3; i++)
        orc_frame_free(&s->frames[i]);
    free(s->frames);
    free(s->qm);
    free(s);
}

orc_frame *orc_store_frame(orc_store *s, uint32_t stream, uint32_t slot)
{
    return &s->frames[(size_t)stream * 3 + slot];
}

void orc_store_set_quant(orc_store *s, uint32_t stream, const uint8_t intra[64], const uint8_t non_intra[64])
{
    memcpy(s->qm[stream][0], intra, 64);
    memcpy(s->qm[stream][1], non_intra, 64);
}

/* video.go:719-744 for one coefficient already placed at natural index idx. */
static int64_t dequant_premult(int64_t level, int intra, int qscale, int qm, int idx)
{
    level *= 2;
    if (!intra)
        level += level < 0 ? -1 : 1;
    level = (level * qscale * qm) >> 4;
    if ((level & 1) == 0)
        level -= level > 0 ? 1 : -1;
    if (level > 2047)
        level = 2047;
    else if (level < -2048)
        level = -2048;
    return level * (int64_t)kd_premult[idx];
}

static int reconstruct_mb(orc_store *s, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mb, const uint8_t *coefs)
{
    orc_frame *cur = orc_store_frame(s, pic->stream, pic->cur);
    int intra = (mb->flags & MPEGHIP_MB_INTRA) != 0;
    if (!intra) { /* predictMacroblock, video.go:608-637: exactly one surviving reference */
        const orc_frame *ref = orc_store_frame(s, pic->stream, (mb->flags & MPEGHIP_MB_REF_BWD) ? pic->bwd : pic->fwd);
        if (orc_copy_macroblock(mb->mv_x, mb->mv_y, mb->mb_y, mb->mb_x, ref, cur) != 0)
            return -1;
    }
    const uint8_t *cp = coefs + (size_t)mb->coef_off * MPEGHIP_COEF_UNIT;
    int raw = (mb->flags & MPEGHIP_MB_COEF_RAW) != 0;
    const uint8_t *qm = s->qm[pic->stream][intra ? 0 : 1];
    for (int b = 0; b < 6; b++) {
        if (!(mb->cbp & (0x20 >> b)))
            continue;
        int64_t block[64];
        if (raw) {
            const int32_t *c = (const int32_t *)cp;
            for (int col = 0; col < 8; col++)
                for (int row = 0; row < 8; row++)
                    block[row * 8 + col] = c[col * 8 + row];
            cp += 2 * MPEGHIP_COEF_UNIT;
        } else {
            const int16_t *c = (const int16_t *)cp;
            for (int col = 0; col < 8; col++) {
                for (int row = 0; row < 8; row++) {
                    int idx = row * 8 + col;
                    int64_t q = c[col * 8 + row];
                    if (intra && idx == 0)
                        block[0] = q * 256; /* video.go:672 */
                    else
                        block[idx] = q ? dequant_premult(q, intra, mb->qscale, qm[idx], idx) : 0;
                }
            }
            cp += MPEGHIP_COEF_UNIT;
        }
        /* n as decodeBlock would have left it: one past the last coefficient in scan order */
        int n = 0;
        for (int k = 63; k >= 0; k--) {
            if (block[kd_zigzag[k]] != 0) {
                n = k + 1;
                break;
            }
        }
        if (n == 0)
            n = 1; /* a coded block holds at least one coefficient; all-zero DC behaves like n==1 */

        uint8_t *d; /* video.go:747-770 */
        int di, scan, lw = cur->luma_w;
        if (b < 4) {
            d = cur->y;
            di = (mb->mb_y * lw + mb->mb_x) << 4;
            scan = lw - 8;
            if (b & 1)
                di += 8;
            if (b & 2)
                di += lw << 3;
        } else {
            d = b == 4 ? cur->cb : cur->cr;
            di = ((mb->mb_y * lw) << 2) + (mb->mb_x << 3);
            scan = (lw >> 1) - 8;
        }
        if (n == 1) { /* video.go:774-777 / 787-790 */
            int64_t value = (block[0] + 128) >> 8;
            if (intra)
                orc_copy_value_to_dest(value, d, di, scan);
            else
                orc_add_value_to_dest(value, d, di, scan);
        } else {
            orc_idct(block, n);
            if (intra)
                orc_copy_block_to_dest(block, d, di, scan);
            else
                orc_add_block_to_dest(block, d, di, scan);
        }
    }
    return 0;
}

typedef struct {
    orc_store *s;
    const mpeghip_pic_desc *pics;
    uint32_t pic0, pic1;
    const mpeghip_mb_desc *mbs;
    const uint8_t *coefs;
    int rc;
} job;

static void *job_run(void *p)
{
    job *j = (job *)p;
    for (uint32_t i = j->pic0; i < j->pic1; i++) {
        const mpeghip_pic_desc *pic = &j->pics[i];
        for (uint32_t m = 0; m < pic->mb_count; m++)
            if (reconstruct_mb(j->s, pic, &j->mbs[pic->mb_first + m], j->coefs) != 0)
                j->rc = -1;
    }
    return NULL;
}

int orc_store_submit(orc_store *s, const mpeghip_pic_desc *pics, uint32_t n_pics,
                     const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                     const void *coefs, size_t coef_bytes, int n_threads)
{
    (void)n_mbs;
    (void)coef_bytes;
    if (n_threads < 1)
        n_threads = 1;
    if ((uint32_t)n_threads > n_pics)
        n_threads = (int)n_pics ? (int)n_pics : 1;
    job *jobs = (job *)calloc((size_t)n_threads, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    int rc = 0;
    for (int t = 0; t < n_threads; t++) {
        jobs[t].s = s;
        jobs[t].pics = pics;
        jobs[t].pic0 = (uint32_t)((uint64_t)n_pics * (uint64_t)t / (uint64_t)n_threads);
        jobs[t].pic1 = (uint32_t)((uint64_t)n_pics * (uint64_t)(t + 1) / (uint64_t)n_threads);
        jobs[t].mbs = mbs;
        jobs[t].coefs = (const uint8_t *)coefs;
        if (n_threads == 1)
            job_run(&jobs[t]);
        else
            pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1)
            pthread_join(th[t], NULL);
        if (jobs[t].rc)
            rc = -1;
    }
    free(jobs);
    free(th);
    return rc;
}

void orc_synth_frames(orc_synth *st, const int32_t *samples, uint32_t n_frames, int format, int fma, void *out)
{ /* audio.go:378-422 */
    static float d[1024];
    static int d_ready;
    if (!d_ready) {
        orc_window_table(d);
        d_ready = 1;
    }
    float u[32];
    int64_t s3[32][3];
    for (uint32_t f = 0; f < n_frames; f++) {
        const int32_t *in = samples + (size_t)f * MPEGHIP_AUDIO_FRAME_INTS;
        int out_pos = 0;
        for (int t = 0; t < 36; t++) {
            st->vpos = (st->vpos - 64) & 1023;
            for (int ch = 0; ch < 2; ch++) {
                for (int sb = 0; sb < 32; sb++) {
                    s3[sb][0] = in[(ch * 36 + t) * 32 + sb];
                    s3[sb][1] = s3[sb][2] = 0;
                }
                orc_idct36((const int64_t(*)[3])s3, 0, st->v[ch], st->vpos);
                orc_synth_window(u, d, st->v[ch], st->vpos, fma);
                for (int j = 0; j < 32; j++) {
                    float sN = u[j] / -1090519040.0f; /* audio.go:390 */
                    size_t il = (size_t)f * 2304 + (size_t)((out_pos + j) << 1) + (size_t)ch;
                    switch (format) {
                    case MPEGHIP_AUDIO_F32N:
                        ((float *)out)[il] = sN;
                        break;
                    case MPEGHIP_AUDIO_F32NLR: /* audio.go:392-399: Left / Right planes */
                        ((float *)out)[(size_t)f * 2304 + (size_t)ch * 1152 + (size_t)(out_pos + j)] = sN;
                        break;
                    case MPEGHIP_AUDIO_S16: /* audio.go:400-408; amd64 float->int16: truncate, keep low 16 bits */
                        ((int16_t *)out)[il] = (int16_t)(int32_t)(sN < 0 ? sN * 32768.0f : sN * 32767.0f);
                        break;
                    default: /* audio.go:409-417: both constants round to 2^31 as float32 */
                        ((float *)out)[il] = sN * 2147483648.0f;
                        break;
                    }
                }
            }
            out_pos += 32;
        }
    }
}
