/*
 * mpeg_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see mpeg_oracle.h).
 *
 * Plain-C restatement of gen2brain/mpeg's CPU decode path.  Go semantics kept:
 * `int` is 64 bit, `>>` on signed values is arithmetic, `/` and `%` truncate
 * toward zero, byte(x) wraps, every float32 operation rounds once (build with
 * -ffp-contract=off), float constants are rounded once from decimal.
 */
#include "mpeg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "iso11172_synth_window.h"
#include "iso11172_vlc_codes.h"

/* ===================================================================== utils */

uint64_t orc_fnv1a64(uint64_t h, const void *data, size_t n)
{
    /* hash/fnv New64a as used by mpeg_test.go:176,217 */
    const uint8_t *p = (const uint8_t *)data;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

static inline int64_t i64abs(int64_t x) { return x < 0 ? -x : x; }

/* ================================================================ bit reader */
/* buffer.go:203-302, 341-376 over a complete in-memory stream.  The reference
 * pulls 128 KiB chunks through a load callback; with the whole stream resident
 * `has` fails exactly when the file is exhausted, which is when the reference
 * sets hasEnded (buffer.go:142-152, 216-218). */
typedef struct {
    const uint8_t *data;
    size_t len;  /* bytes */
    size_t bit;  /* bitIndex */
    int ended;
} orc_bits;

static int bits_has(orc_bits *b, size_t count)
{
    if (b->len * 8 >= b->bit && b->len * 8 - b->bit >= count)
        return 1;
    b->ended = 1;
    return 0;
}

/* buffer.go:246-255; reading past the end panics in Go — here it yields 0 bits. */
static int bits_read1(orc_bits *b)
{
    size_t byte = b->bit >> 3;
    int v = 0;
    if (byte < b->len)
        v = (b->data[byte] >> (7 - (b->bit & 7))) & 1;
    b->bit++;
    return v;
}

/* buffer.go:223-244 */
static int64_t bits_read(orc_bits *b, int count)
{
    int64_t v = 0;
    while (count-- > 0)
        v = (v << 1) | bits_read1(b);
    return v;
}

static void bits_align(orc_bits *b) { b->bit = ((b->bit + 7) >> 3) << 3; } /* buffer.go:257-259 */

static void bits_skip(orc_bits *b, size_t count) /* buffer.go:261-265 */
{
    if (bits_has(b, count))
        b->bit += count;
}

static int bits_skip_bytes(orc_bits *b, uint8_t v) /* buffer.go:267-277 */
{
    bits_align(b);
    int skipped = 0;
    while (bits_has(b, 8) && b->data[b->bit >> 3] == v) {
        b->bit += 8;
        skipped++;
    }
    return skipped;
}

static int bits_next_start_code(orc_bits *b) /* buffer.go:279-302 */
{
    bits_align(b);
    while (b->len * 8 >= b->bit + 40) {
        size_t i = b->bit >> 3;
        if (b->data[i] == 0 && b->data[i + 1] == 0 && b->data[i + 2] == 1) {
            b->bit = (i + 4) << 3;
            return b->data[i + 3];
        }
        b->bit += 8;
    }
    b->ended = 1; /* has(40) failed with the source exhausted */
    return -1;
}

static int bits_find_start_code(orc_bits *b, int code) /* buffer.go:304-311 */
{
    for (;;) {
        int cur = bits_next_start_code(b);
        if (cur == code || cur == -1)
            return cur;
    }
}

static int bits_has_start_code(orc_bits *b, int code) /* buffer.go:313-324 */
{
    size_t prev = b->bit;
    int cur = bits_find_start_code(b, code);
    b->bit = prev;
    return cur;
}

static int bits_peek_non_zero(orc_bits *b, int count) /* buffer.go:341-350 */
{
    if (!bits_has(b, (size_t)count))
        return 0;
    int64_t v = bits_read(b, count);
    b->bit -= (size_t)count;
    return v != 0;
}

/* buffer.go:352-376 — bit-by-bit walk of a binary code tree.  The tree is
 * built once from the ISO code lists; a "dead" prefix ends the walk with 0. */
typedef struct { int16_t next[2]; int32_t value; int8_t leaf[2]; } vlc_node;
typedef struct { vlc_node *n; int count, cap; } vlc_tree;

static void vlc_build(vlc_tree *t, const orc_vlc_code *codes)
{
    t->cap = 512;
    t->n = (vlc_node *)calloc((size_t)t->cap, sizeof(vlc_node) * 2);
    t->count = 1;
    /* node i: child c is either a leaf (value in leafval[i][c]) or an inner node */
    for (const orc_vlc_code *c = codes; c->bits; c++) {
        int node = 0;
        size_t L = strlen(c->bits);
        for (size_t k = 0; k < L; k++) {
            int bit = c->bits[k] - '0';
            if (k + 1 == L) {
                t->n[node].leaf[bit] = 1;
                /* store the leaf value in a side node */
                int leafnode = t->count++;
                t->n[node].next[bit] = (int16_t)leafnode;
                t->n[leafnode].value = c->dead ? 0 : c->value;
            } else {
                if (!t->n[node].next[bit]) {
                    t->n[node].next[bit] = (int16_t)t->count++;
                }
                node = t->n[node].next[bit];
            }
        }
    }
}

static int vlc_read(orc_bits *b, const vlc_tree *t)
{
    int node = 0;
    for (;;) {
        int bit = bits_read1(b);
        int nxt = t->n[node].next[bit];
        if (t->n[node].leaf[bit])
            return t->n[nxt].value;
        if (!nxt)
            return 0; /* cannot happen: every prefix is covered by a code or a dead end */
        node = nxt;
    }
}

static vlc_tree T_mba, T_mbtype[4], T_cbp, T_motion, T_dcsize[3], T_coeff;
static int tables_ready;

static void tables_init(void)
{
    if (tables_ready)
        return;
    vlc_build(&T_mba, orc_vlc_mba_increment);
    vlc_build(&T_mbtype[1], orc_vlc_mb_type_i);
    vlc_build(&T_mbtype[2], orc_vlc_mb_type_p);
    vlc_build(&T_mbtype[3], orc_vlc_mb_type_b);
    vlc_build(&T_cbp, orc_vlc_coded_block_pattern);
    vlc_build(&T_motion, orc_vlc_motion_code);
    vlc_build(&T_dcsize[0], orc_vlc_dct_dc_size_luma);
    vlc_build(&T_dcsize[1], orc_vlc_dct_dc_size_chroma);
    T_dcsize[2] = T_dcsize[1];
    vlc_build(&T_coeff, orc_vlc_dct_coeff);
    tables_ready = 1;
}

/* =============================================================== video: data */

static const uint8_t k_zigzag[64] = { /* video.go:1044-1053 (ISO 11172-2 fig. 2-D.30 scan) */
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static const uint8_t k_intra_q[64] = { /* video.go:1055-1064 (ISO default intra matrix) */
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};

static const uint8_t k_premult[64] = { /* video.go:1077-1086 */
    32, 44, 42, 38, 32, 25, 17, 9,  44, 62, 58, 52, 44, 35, 24, 12,
    42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
    32, 44, 42, 38, 32, 25, 17, 9,  25, 35, 33, 30, 25, 20, 14, 7,
    17, 24, 23, 20, 17, 14, 9,  5,  9,  12, 12, 10, 9,  7,  5,  2};

static const double k_picture_rate[16] = { /* video.go:1034-1037 */
    0.000, 23.976, 24.000, 25.000, 29.970, 30.000, 50.000, 59.940,
    60.000, 0, 0, 0, 0, 0, 0, 0};

enum { PIC_I = 1, PIC_P = 2, PIC_B = 3 };
enum { START_PICTURE = 0x00, START_SLICE_FIRST = 0x01, START_SLICE_LAST = 0xAF,
       START_USER_DATA = 0xB2, START_SEQUENCE = 0xB3, START_EXTENSION = 0xB5 };

/* ============================================================ video: kernels */

static _Thread_local int64_t g_idct_mid; /* running max |intermediate| of the last orc_idct call */
#define TRACK(x) do { int64_t a_ = i64abs(x); if (a_ > g_idct_mid) g_idct_mid = a_; } while (0)

/* One 8-point pass of video.go:870-895 (columns) / :900-925 (rows). */
static void idct_1d(const int64_t in[8], int64_t out[8], int final_shift)
{
    int64_t b1 = in[4];
    int64_t b3 = in[2] + in[6];
    int64_t b4 = in[5] - in[3];
    int64_t tmp1 = in[1] + in[7];
    int64_t tmp2 = in[3] + in[5];
    int64_t b6 = in[1] - in[7];
    int64_t b7 = tmp1 + tmp2;
    int64_t m0 = in[0];
    int64_t p1 = b6 * 473 - b4 * 196 + 128;
    int64_t x4 = (p1 >> 8) - b7;
    int64_t p2 = (tmp1 - tmp2) * 362 + 128;
    int64_t x0 = x4 - (p2 >> 8);
    int64_t x1 = m0 - b1;
    int64_t p3 = (in[2] - in[6]) * 362 + 128;
    int64_t x2 = (p3 >> 8) - b3;
    int64_t x3 = m0 + b1;
    int64_t y3 = x1 + x2;
    int64_t y4 = x3 + b3;
    int64_t y5 = x1 - x2;
    int64_t y6 = x3 - b3;
    int64_t p4 = b4 * 473 + b6 * 196 + 128;
    int64_t y7 = -x0 - (p4 >> 8);
    TRACK(b6 * 473); TRACK(b4 * 196); TRACK(p1); TRACK(p2); TRACK(p3); TRACK(b4 * 473); TRACK(b6 * 196); TRACK(p4);
    TRACK(b7); TRACK(x4); TRACK(x0); TRACK(x1); TRACK(x2); TRACK(x3); TRACK(y3); TRACK(y4); TRACK(y5); TRACK(y6); TRACK(y7);
    out[0] = b7 + y4;
    out[1] = x4 + y3;
    out[2] = y5 - x0;
    out[3] = y6 - y7;
    out[4] = y6 + y7;
    out[5] = x0 + y5;
    out[6] = y3 - x4;
    out[7] = y4 - b7;
    for (int k = 0; k < 8; k++) {
        TRACK(out[k]);
        if (final_shift)
            out[k] = (out[k] + 128) >> 8;
    }
}

int64_t orc_idct(int64_t block[64], int max_index)
{
    int64_t in[8], out[8];
    g_idct_mid = 0;
    if (max_index < 10) {
        /* video.go:807-866: only rows 0-3 of columns 0-3 are read, rows of
         * columns 4-7 are treated as zero, stale values there are overwritten. */
        for (int i = 0; i < 4; i++) {
            for (int r = 0; r < 8; r++)
                in[r] = r < 4 ? block[r * 8 + i] : 0;
            idct_1d(in, out, 0);
            for (int r = 0; r < 8; r++)
                block[r * 8 + i] = out[r];
        }
        for (int i = 0; i < 64; i += 8) {
            for (int c = 0; c < 8; c++)
                in[c] = c < 4 ? block[i + c] : 0;
            idct_1d(in, out, 1);
            for (int c = 0; c < 8; c++)
                block[i + c] = out[c];
        }
    } else {
        for (int i = 0; i < 8; i++) { /* video.go:869-896 */
            for (int r = 0; r < 8; r++)
                in[r] = block[r * 8 + i];
            idct_1d(in, out, 0);
            for (int r = 0; r < 8; r++)
                block[r * 8 + i] = out[r];
        }
        for (int i = 0; i < 64; i += 8) { /* video.go:899-926 */
            idct_1d(&block[i], out, 1);
            for (int c = 0; c < 8; c++)
                block[i + c] = out[c];
        }
    }
    return g_idct_mid;
}

static inline uint8_t clamp_u8(int64_t n) /* video.go:1014-1016 */
{
    return (uint8_t)(n < 0 ? 0 : (n > 255 ? 255 : n));
}

void orc_copy_block_to_dest(const int64_t block[64], uint8_t *dest, int index, int scan)
{ /* video.go:943-956 */
    for (int n = 0; n < 64; n += 8) {
        for (int k = 0; k < 8; k++)
            dest[index + k] = clamp_u8(block[n + k]);
        index += scan + 8;
    }
}

void orc_add_block_to_dest(const int64_t block[64], uint8_t *dest, int index, int scan)
{ /* video.go:958-971 */
    for (int n = 0; n < 64; n += 8) {
        for (int k = 0; k < 8; k++)
            dest[index + k] = clamp_u8((int64_t)dest[index + k] + block[n + k]);
        index += scan + 8;
    }
}

void orc_copy_value_to_dest(int64_t value, uint8_t *dest, int index, int scan)
{ /* video.go:973-987 */
    uint8_t val = clamp_u8(value);
    for (int n = 0; n < 64; n += 8) {
        memset(dest + index, val, 8);
        index += scan + 8;
    }
}

void orc_add_value_to_dest(int64_t value, uint8_t *dest, int index, int scan)
{ /* video.go:989-1002 */
    for (int n = 0; n < 64; n += 8) {
        for (int k = 0; k < 8; k++)
            dest[index + k] = clamp_u8((int64_t)dest[index + k] + value);
        index += scan + 8;
    }
}

int orc_frame_alloc(orc_frame *f, int width, int height)
{ /* video.go:314-322, 333-355 */
    memset(f, 0, sizeof(*f));
    int mb_w = (width + 15) >> 4, mb_h = (height + 15) >> 4;
    f->width = width;
    f->height = height;
    f->luma_w = mb_w << 4;
    f->luma_h = mb_h << 4;
    f->chroma_w = mb_w << 3;
    f->chroma_h = mb_h << 3;
    f->luma_size = (size_t)f->luma_w * (size_t)f->luma_h;
    f->chroma_size = (size_t)f->chroma_w * (size_t)f->chroma_h;
    f->total = f->luma_size + 2 * f->chroma_size + (size_t)f->luma_w * 16;
    f->base = (uint8_t *)calloc(f->total, 1);
    if (!f->base)
        return -1;
    f->y = f->base;
    f->cb = f->base + f->luma_size;
    f->cr = f->cb + f->chroma_size;
    return 0;
}

void orc_frame_free(orc_frame *f)
{
    free(f->base);
    memset(f, 0, sizeof(*f));
}

/* video_noasm.go:7-26 */
#define LO_BYTE_MASK 0x00ff00ff00ff00ffull
#define AVG_MASK     0x7f7f7f7f7f7f7f7full
#define TWO_PER_LANE 0x0002000200020002ull

static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void st64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
static inline uint64_t round_avg(uint64_t a, uint64_t b) { return (a | b) - (((a ^ b) >> 1) & AVG_MASK); }
static inline uint64_t bilin_avg(uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
    uint64_t lo = (((a & LO_BYTE_MASK) + (b & LO_BYTE_MASK) + (c & LO_BYTE_MASK) + (d & LO_BYTE_MASK) + TWO_PER_LANE) >> 2) & LO_BYTE_MASK;
    uint64_t hi = ((((a >> 8) & LO_BYTE_MASK) + ((b >> 8) & LO_BYTE_MASK) + ((c >> 8) & LO_BYTE_MASK) + ((d >> 8) & LO_BYTE_MASK) + TWO_PER_LANE) >> 2) & LO_BYTE_MASK;
    return lo | (hi << 8);
}

/* Extent check of one copyBlock call: Go indexes src[:cap(src)] so the legal
 * range is [plane start, end of base); outside it the reference panics.
 * *over receives 1 if the block ends beyond its own plane (legal, "over-read"). */
static int block_in_range(const orc_frame *s, const uint8_t *plane, size_t plane_len,
                          int stride, int64_t si, int size, int odd_h, int odd_v, int *over)
{
    int64_t cap = (int64_t)(s->base + s->total - plane);
    int64_t last = si + (int64_t)(size - 1 + (odd_v ? 1 : 0)) * stride + size - 1 + (odd_h ? 1 : 0);
    if (si < 0 || last >= cap)
        return 0;
    if (over && last >= (int64_t)plane_len)
        *over = 1;
    return 1;
}

/* video_noasm.go:48-80 */
static void copy_block(const uint8_t *src, uint8_t *dst, int stride, int64_t si, int64_t di,
                       int size, int odd_h, int odd_v)
{
    for (int r = 0; r < size; r++) {
        if (!odd_h && !odd_v) {
            memcpy(dst + di, src + si, (size_t)size);
        } else if (odd_h && !odd_v) {
            for (int x = 0; x < size; x += 8)
                st64(dst + di + x, round_avg(ld64(src + si + x), ld64(src + si + x + 1)));
        } else if (!odd_h && odd_v) {
            for (int x = 0; x < size; x += 8)
                st64(dst + di + x, round_avg(ld64(src + si + x), ld64(src + si + x + stride)));
        } else {
            for (int x = 0; x < size; x += 8)
                st64(dst + di + x, bilin_avg(ld64(src + si + x), ld64(src + si + x + 1),
                                             ld64(src + si + x + stride), ld64(src + si + x + stride + 1)));
        }
        si += stride;
        di += stride;
    }
}

static _Thread_local int g_last_overread;

/* would this copyMacroblock call stay inside the reference's slices (1), or panic (0)? */
static int copy_macroblock_in_range(int motion_h, int motion_v, int mb_row, int mb_col, const orc_frame *s)
{
    int lw = s->luma_w, cw = s->chroma_w;
    int64_t lsi = ((int64_t)(mb_row << 4) + (motion_v >> 1)) * lw + (mb_col << 4) + (motion_h >> 1);
    int cm_h = motion_h / 2, cm_v = motion_v / 2;
    int64_t csi = ((int64_t)(mb_row << 3) + (cm_v >> 1)) * cw + (mb_col << 3) + (cm_h >> 1);
    return block_in_range(s, s->y, s->luma_size, lw, lsi, 16, (motion_h & 1) == 1, (motion_v & 1) == 1, NULL) &&
           block_in_range(s, s->cb, s->chroma_size, cw, csi, 8, (cm_h & 1) == 1, (cm_v & 1) == 1, NULL) &&
           block_in_range(s, s->cr, s->chroma_size, cw, csi, 8, (cm_h & 1) == 1, (cm_v & 1) == 1, NULL);
}

int orc_copy_macroblock(int motion_h, int motion_v, int mb_row, int mb_col,
                        const orc_frame *s, orc_frame *d)
{ /* video_noasm.go:28-43 */
    int lw = s->luma_w, cw = s->chroma_w;
    int hp = motion_h >> 1, vp = motion_v >> 1;
    int64_t lsi = ((int64_t)(mb_row << 4) + vp) * lw + (mb_col << 4) + hp;
    int64_t ldi = (int64_t)(mb_row << 4) * lw + (mb_col << 4);
    int l_oh = (motion_h & 1) == 1, l_ov = (motion_v & 1) == 1;

    int cm_h = motion_h / 2, cm_v = motion_v / 2; /* truncation toward zero */
    int chp = cm_h >> 1, cvp = cm_v >> 1;
    int64_t csi = ((int64_t)(mb_row << 3) + cvp) * cw + (mb_col << 3) + chp;
    int64_t cdi = (int64_t)(mb_row << 3) * cw + (mb_col << 3);
    int c_oh = (cm_h & 1) == 1, c_ov = (cm_v & 1) == 1;

    int over = 0;
    if (!block_in_range(s, s->y, s->luma_size, lw, lsi, 16, l_oh, l_ov, &over) ||
        !block_in_range(s, s->cb, s->chroma_size, cw, csi, 8, c_oh, c_ov, &over) ||
        !block_in_range(s, s->cr, s->chroma_size, cw, csi, 8, c_oh, c_ov, &over))
        return -1;
    g_last_overread = over;
    copy_block(s->y, d->y, lw, lsi, ldi, 16, l_oh, l_ov);
    copy_block(s->cb, d->cb, cw, csi, cdi, 8, c_oh, c_ov);
    copy_block(s->cr, d->cr, cw, csi, cdi, 8, c_oh, c_ov);
    return 0;
}

int orc_copy_macroblock_ref(int motion_h, int motion_v, int mb_row, int mb_col,
                            const orc_frame *s, orc_frame *d)
{ /* video_test.go:10-43 */
    const uint8_t *sp[3] = {s->y, s->cb, s->cr};
    uint8_t *dp[3] = {d->y, d->cb, d->cr};
    for (int p = 0; p < 3; p++) {
        int size = p ? 8 : 16, stride = p ? s->chroma_w : s->luma_w;
        int mh = p ? motion_h / 2 : motion_h, mv = p ? motion_v / 2 : motion_v;
        int hp = mh >> 1, vp = mv >> 1, oh = (mh & 1) == 1, ov = (mv & 1) == 1;
        for (int y = 0; y < size; y++) {
            for (int x = 0; x < size; x++) {
                int64_t si = ((int64_t)(mb_row * size) + vp + y) * stride + (mb_col * size) + hp + x;
                int64_t di = ((int64_t)(mb_row * size) + y) * stride + (mb_col * size) + x;
                const uint8_t *q = sp[p] + si;
                int v;
                if (!oh && !ov)      v = q[0];
                else if (oh && !ov)  v = (q[0] + q[1] + 1) >> 1;
                else if (!oh && ov)  v = (q[0] + q[stride] + 1) >> 1;
                else                 v = (q[0] + q[1] + q[stride] + q[stride + 1] + 2) >> 2;
                dp[p][di] = (uint8_t)v;
            }
        }
    }
    return 0;
}

void orc_test_frame_fill(orc_frame *f, int fill)
{ /* video_test.go:45-59: square lumaWidth x lumaWidth planes, no pad */
    for (size_t i = 0; i < f->luma_size; i++)
        f->y[i] = (uint8_t)(((int64_t)i * 131 + fill * 7) & 0xff);
    for (size_t i = 0; i < f->chroma_size; i++) {
        f->cb[i] = (uint8_t)(((int64_t)i * 197 + fill * 13) & 0xff);
        f->cr[i] = (uint8_t)(((int64_t)i * 251 + fill * 29) & 0xff);
    }
}

void orc_ycbcr_to_rgba(const orc_frame *f, uint8_t *rgba)
{
    /* video.go:31-36 -> draw.Draw(dst *image.RGBA, r, src *image.YCbCr, sp, draw.Src)
     * -> imageutil.DrawYCbCr (Go 1.23 standard library, not part of the reference
     * tree), 4:2:0 branch; YStride=luma_w, CStride=chroma_w (video.go:357-365). */
    for (int y = 0; y < f->height; y++) {
        const uint8_t *yr = f->y + (size_t)y * f->luma_w;
        const uint8_t *cbr = f->cb + (size_t)(y / 2) * f->chroma_w;
        const uint8_t *crr = f->cr + (size_t)(y / 2) * f->chroma_w;
        uint8_t *o = rgba + (size_t)y * f->width * 4;
        for (int x = 0; x < f->width; x++) {
            int32_t yy1 = (int32_t)yr[x] * 0x10101;
            int32_t cb1 = (int32_t)cbr[x / 2] - 128;
            int32_t cr1 = (int32_t)crr[x / 2] - 128;
            int32_t r = yy1 + 91881 * cr1;
            if (((uint32_t)r & 0xff000000u) == 0) r >>= 16; else r = ~(r >> 31);
            int32_t g = yy1 - 22554 * cb1 - 46802 * cr1;
            if (((uint32_t)g & 0xff000000u) == 0) g >>= 16; else g = ~(g >> 31);
            int32_t b = yy1 + 116130 * cb1;
            if (((uint32_t)b & 0xff000000u) == 0) b >>= 16; else b = ~(b >> 31);
            o[4 * x + 0] = (uint8_t)r;
            o[4 * x + 1] = (uint8_t)g;
            o[4 * x + 2] = (uint8_t)b;
            o[4 * x + 3] = 255;
        }
    }
}

/* ============================================================ video: decoder */

typedef struct { int full_px, r_size, h, v, is_set; } orc_motion; /* video.go:1026-1032 */

struct orc_video {
    orc_bits buf;
    double frame_rate, time;
    int frames_decoded;
    int width, height, mb_w, mb_h, mb_size;
    int luma_w, luma_h, chroma_w, chroma_h;
    int start_code, picture_type;
    orc_motion mf, mb;
    int has_seq;
    int qscale, slice_begin, mb_addr, mb_row, mb_col, mb_type, mb_intra;
    int mb_dropped; /* a copyMacroblock call of this macroblock would panic in the reference: nothing of it is written */
    int64_t dc_pred[3];
    orc_frame cur, fwd, bwd;
    int64_t block[64];
    uint8_t iq[64], niq[64];
    int has_ref, no_delay;
    orc_video_stats st;
};

static int decode_sequence_header(orc_video *v)
{ /* video.go:270-331 */
    orc_bits *b = &v->buf;
    if (!bits_has(b, 64 + 2 * 64 * 8))
        return 0;
    v->width = (int)bits_read(b, 12);
    v->height = (int)bits_read(b, 12);
    if (v->width <= 0 || v->height <= 0)
        return 0;
    bits_read(b, 4); /* aspect ratio */
    v->frame_rate = k_picture_rate[bits_read(b, 4)];
    bits_read(b, 18); /* bit rate */
    bits_skip(b, 1 + 10 + 1);
    if (bits_read1(b)) {
        for (int i = 0; i < 64; i++)
            v->iq[k_zigzag[i]] = (uint8_t)bits_read(b, 8);
    } else {
        memcpy(v->iq, k_intra_q, 64);
    }
    if (bits_read1(b)) {
        for (int i = 0; i < 64; i++)
            v->niq[k_zigzag[i]] = (uint8_t)bits_read(b, 8);
    } else {
        memset(v->niq, 16, 64); /* video.go:1066-1075 */
    }
    v->mb_w = (v->width + 15) >> 4;
    v->mb_h = (v->height + 15) >> 4;
    v->mb_size = v->mb_w * v->mb_h;
    v->luma_w = v->mb_w << 4;
    v->luma_h = v->mb_h << 4;
    v->chroma_w = v->mb_w << 3;
    v->chroma_h = v->mb_h << 3;
    orc_frame_free(&v->cur);
    orc_frame_free(&v->fwd);
    orc_frame_free(&v->bwd);
    if (orc_frame_alloc(&v->cur, v->width, v->height) || orc_frame_alloc(&v->fwd, v->width, v->height) ||
        orc_frame_alloc(&v->bwd, v->width, v->height))
        return 0;
    v->has_seq = 1;
    return 1;
}

static void copy_mb_counted(orc_video *v, int mh, int mv, const orc_frame *s)
{
    v->st.copy_mb_calls++;
    v->st.copy_mode[(mh & 1) | ((mv & 1) << 1)]++;
    if (abs(mh) > v->st.max_abs_mv) v->st.max_abs_mv = abs(mh);
    if (abs(mv) > v->st.max_abs_mv) v->st.max_abs_mv = abs(mv);
    if (orc_copy_macroblock(mh, mv, v->mb_row, v->mb_col, s, &v->cur) != 0)
        v->st.range_errors++;
    else if (g_last_overread)
        v->st.overreads++;
}

static void predict_macroblock(orc_video *v)
{ /* video.go:608-637 */
    int fw_h = v->mf.h, fw_v = v->mf.v;
    if (v->mf.full_px) {
        fw_h <<= 1;
        fw_v <<= 1;
    }
    /* Where the reference would panic (a source slice out of range, video_noasm.go:48-50) there is nothing to restate: this
     * build defines that the WHOLE macroblock is dropped — none of its copies, none of its blocks reach the frame — and the
     * parse goes on.  All copies the reference would make are checked before the first one is made. */
    {
        int bw_h = v->mb.h * (v->mb.full_px ? 2 : 1), bw_v = v->mb.v * (v->mb.full_px ? 2 : 1);
        int ok = 1;
        if (v->picture_type == PIC_B) {
            if (v->mf.is_set)
                ok = copy_macroblock_in_range(fw_h, fw_v, v->mb_row, v->mb_col, &v->fwd) &&
                     (!v->mb.is_set || copy_macroblock_in_range(bw_h, bw_v, v->mb_row, v->mb_col, &v->bwd));
            else
                ok = copy_macroblock_in_range(bw_h, bw_v, v->mb_row, v->mb_col, &v->bwd);
        } else {
            ok = copy_macroblock_in_range(fw_h, fw_v, v->mb_row, v->mb_col, &v->fwd);
        }
        if (!ok) {
            v->st.range_errors++;
            v->mb_dropped = 1;
            return;
        }
    }
    if (v->picture_type == PIC_B) {
        int bw_h = v->mb.h, bw_v = v->mb.v;
        if (v->mb.full_px) {
            bw_h <<= 1;
            bw_v <<= 1;
        }
        if (v->mf.is_set) {
            copy_mb_counted(v, fw_h, fw_v, &v->fwd);
            if (v->mb.is_set) {
                v->st.bidir_mbs++;
                copy_mb_counted(v, bw_h, bw_v, &v->bwd); /* overwrites, never averages */
            }
        } else {
            copy_mb_counted(v, bw_h, bw_v, &v->bwd);
        }
    } else {
        copy_mb_counted(v, fw_h, fw_v, &v->fwd);
    }
}

static int decode_motion_vector(orc_video *v, int r_size, int motion)
{ /* video.go:583-606 */
    int fscale = 1 << r_size;
    int m_code = vlc_read(&v->buf, &T_motion);
    int d;
    if (m_code != 0 && fscale != 1) {
        int r = (int)bits_read(&v->buf, r_size);
        d = ((abs(m_code) - 1) << r_size) + r + 1;
        if (m_code < 0)
            d = -d;
    } else {
        d = m_code;
    }
    motion += d;
    if (motion > (fscale << 4) - 1)
        motion -= fscale << 5;
    else if (motion < ((-fscale) << 4))
        motion += fscale << 5;
    return motion;
}

static void decode_motion_vectors(orc_video *v)
{ /* video.go:564-581 */
    if (v->mf.is_set) {
        v->mf.h = decode_motion_vector(v, v->mf.r_size, v->mf.h);
        v->mf.v = decode_motion_vector(v, v->mf.r_size, v->mf.v);
    } else if (v->picture_type == PIC_P) {
        v->mf.h = 0;
        v->mf.v = 0;
    }
    if (v->mb.is_set) {
        v->mb.h = decode_motion_vector(v, v->mb.r_size, v->mb.h);
        v->mb.v = decode_motion_vector(v, v->mb.r_size, v->mb.v);
    }
}

static void decode_block(orc_video *v, int block)
{ /* video.go:639-799 */
    orc_bits *b = &v->buf;
    int n = 0;
    const uint8_t *qm;

    if (v->mb_intra) {
        int plane = block > 3 ? block - 3 : 0;
        int64_t predictor = v->dc_pred[plane];
        int dct_size = vlc_read(b, &T_dcsize[plane]);
        if (dct_size > 0) {
            int64_t differential = bits_read(b, dct_size);
            if (differential & ((int64_t)1 << (dct_size - 1)))
                v->block[0] = predictor + differential;
            else
                v->block[0] = predictor + ((-((int64_t)1 << dct_size)) | (differential + 1));
        } else {
            v->block[0] = predictor;
        }
        v->dc_pred[plane] = v->block[0];
        v->block[0] *= 256; /* <<= 3+5, written as a multiply: shifting a negative value is UB in C */
        qm = v->iq;
        n = 1;
    } else {
        qm = v->niq;
    }

    int64_t level = 0;
    for (;;) {
        int run;
        int coeff = vlc_read(b, &T_coeff);
        if (coeff == 0x0001 && n > 0 && bits_read1(b) == 0)
            break; /* end_of_block */
        if (coeff == 0xffff) {
            run = (int)bits_read(b, 6);
            level = bits_read(b, 8);
            if (level == 0)
                level = bits_read(b, 8);
            else if (level == 128)
                level = bits_read(b, 8) - 256;
            else if (level > 128)
                level -= 256;
        } else {
            run = coeff >> 8;
            level = coeff & 0xff;
            if (bits_read1(b))
                level = -level;
        }
        n += run;
        if (n < 0 || n >= 64) {
            v->st.invalid_blocks++;
            return; /* invalid: blockData keeps whatever was written so far */
        }
        int dz = k_zigzag[n] & 63;
        n++;

        level *= 2;
        if (!v->mb_intra)
            level += level < 0 ? -1 : 1;
        level = (level * v->qscale * (int64_t)qm[dz]) >> 4;
        if ((level & 1) == 0)
            level -= level > 0 ? 1 : -1;
        if (level > 2047)
            level = 2047;
        else if (level < -2048)
            level = -2048;
        v->block[dz] = level * (int64_t)k_premult[dz];
    }

    uint8_t *d;
    int di, scan;
    if (block < 4) {
        d = v->cur.y;
        di = (v->mb_row * v->luma_w + v->mb_col) << 4;
        scan = v->luma_w - 8;
        if (block & 1)
            di += 8;
        if (block & 2)
            di += v->luma_w << 3;
    } else {
        d = block == 4 ? v->cur.cb : v->cur.cr;
        di = ((v->mb_row * v->luma_w) << 2) + (v->mb_col << 3);
        scan = (v->luma_w >> 1) - 8;
    }

    v->st.coded_blocks++;
    if (v->mb_dropped) { /* (see predict_macroblock) the block's bookkeeping goes on, the frame is not touched */
        if (n == 1)
            v->block[0] = 0;
        else
            memset(v->block, 0, sizeof(v->block));
        return;
    }
    if (n == 1) {
        v->st.dc_only_blocks++;
        int64_t value = (v->block[0] + 128) >> 8;
        if (v->mb_intra)
            orc_copy_value_to_dest(value, d, di, scan);
        else
            orc_add_value_to_dest(value, d, di, scan);
        v->block[0] = 0;
    } else {
        for (int i = 0; i < 64; i++)
            if (i64abs(v->block[i]) > v->st.max_idct_in)
                v->st.max_idct_in = i64abs(v->block[i]);
        if (n < 10) v->st.sparse_idct++; else v->st.full_idct++;
        int64_t mid = orc_idct(v->block, n);
        if (mid > v->st.max_idct_mid) v->st.max_idct_mid = mid;
        for (int i = 0; i < 64; i++)
            if (i64abs(v->block[i]) > v->st.max_idct_out)
                v->st.max_idct_out = i64abs(v->block[i]);
        if (v->mb_intra)
            orc_copy_block_to_dest(v->block, d, di, scan);
        else
            orc_add_block_to_dest(v->block, d, di, scan);
        memset(v->block, 0, sizeof(v->block));
    }
}

static void decode_macroblock(orc_video *v)
{ /* video.go:462-562 */
    orc_bits *b = &v->buf;
    int increment = 0;
    int t = vlc_read(b, &T_mba);
    while (t == 34)
        t = vlc_read(b, &T_mba); /* stuffing */
    while (t == 35) {
        increment += 33; /* escape */
        t = vlc_read(b, &T_mba);
    }
    increment += t;

    if (v->slice_begin) {
        v->slice_begin = 0;
        v->mb_addr += increment;
    } else {
        if (v->mb_addr + increment >= v->mb_size)
            return;
        if (increment > 1) {
            v->dc_pred[0] = v->dc_pred[1] = v->dc_pred[2] = 128;
            if (v->picture_type == PIC_P) {
                v->mf.h = 0;
                v->mf.v = 0;
            }
        }
        while (increment > 1) {
            v->mb_addr++;
            v->mb_row = v->mb_addr / v->mb_w;
            v->mb_col = v->mb_addr % v->mb_w;
            v->st.skipped_mbs++;
            v->mb_dropped = 0;
            predict_macroblock(v);
            increment--;
        }
        v->mb_addr++;
    }

    v->mb_row = v->mb_addr / v->mb_w;
    v->mb_col = v->mb_addr % v->mb_w;
    if (v->mb_col >= v->mb_w || v->mb_row >= v->mb_h)
        return;
    if (v->mb_addr < 0)
        return; /* Go would index a negative plane offset and panic; unreachable on the fixtures */

    v->mb_dropped = 0;
    v->mb_type = vlc_read(b, &T_mbtype[v->picture_type]);
    v->mb_intra = (v->mb_type & 0x01) != 0;
    v->mf.is_set = (v->mb_type & 0x08) != 0;
    v->mb.is_set = (v->mb_type & 0x04) != 0;
    if (v->mb_type & 0x10)
        v->qscale = (int)bits_read(b, 5);

    v->st.coded_mbs++;
    if (v->mb_intra) {
        v->st.intra_mbs++;
        v->mf.h = v->mb.h = 0;
        v->mf.v = v->mb.v = 0;
    } else {
        v->dc_pred[0] = v->dc_pred[1] = v->dc_pred[2] = 128;
        decode_motion_vectors(v);
        predict_macroblock(v);
    }

    int cbp = 0;
    if (v->mb_type & 0x02)
        cbp = vlc_read(b, &T_cbp);
    else if (v->mb_intra)
        cbp = 0x3f;
    for (int block = 0, mask = 0x20; block < 6; block++, mask >>= 1)
        if (cbp & mask)
            decode_block(v, block);
}

static void decode_slice(orc_video *v, int slice)
{ /* video.go:436-460 */
    orc_bits *b = &v->buf;
    v->slice_begin = 1;
    v->mb_addr = (slice - 1) * v->mb_w - 1;
    v->mf.h = v->mb.h = 0;
    v->mf.v = v->mb.v = 0;
    v->dc_pred[0] = v->dc_pred[1] = v->dc_pred[2] = 128;
    v->qscale = (int)bits_read(b, 5);
    while (bits_read1(b))
        bits_skip(b, 8);
    do {
        decode_macroblock(v);
    } while (v->mb_addr < v->mb_size - 1 && bits_peek_non_zero(b, 23));
}

static void decode_picture(orc_video *v)
{ /* video.go:374-434 */
    orc_bits *b = &v->buf;
    bits_skip(b, 10);
    v->picture_type = (int)bits_read(b, 3);
    bits_skip(b, 16);
    if (v->picture_type <= 0 || v->picture_type > PIC_B)
        return;
    if (v->picture_type == PIC_P || v->picture_type == PIC_B) {
        v->mf.full_px = bits_read1(b);
        int f_code = (int)bits_read(b, 3);
        if (f_code == 0)
            return;
        v->mf.r_size = f_code - 1;
    }
    if (v->picture_type == PIC_B) {
        v->mb.full_px = bits_read1(b);
        int f_code = (int)bits_read(b, 3);
        if (f_code == 0)
            return;
        v->mb.r_size = f_code - 1;
    }
    v->st.pictures[v->picture_type]++;

    orc_frame frame_temp = v->fwd;
    if (v->picture_type == PIC_I || v->picture_type == PIC_P)
        v->fwd = v->bwd;

    do {
        v->start_code = bits_next_start_code(b);
    } while (v->start_code == START_EXTENSION || v->start_code == START_USER_DATA);

    while (v->start_code >= START_SLICE_FIRST && v->start_code <= START_SLICE_LAST) {
        decode_slice(v, v->start_code & 0xFF);
        if (v->mb_addr >= v->mb_size - 2)
            break;
        v->start_code = bits_next_start_code(b);
    }

    if (v->picture_type == PIC_I || v->picture_type == PIC_P) {
        v->bwd = v->cur;
        v->cur = frame_temp;
    }
}

orc_video *orc_video_open(const uint8_t *data, size_t len)
{ /* video.go:110-121 */
    tables_init();
    orc_video *v = (orc_video *)calloc(1, sizeof(*v));
    if (!v)
        return NULL;
    v->buf.data = data;
    v->buf.len = len;
    v->start_code = bits_find_start_code(&v->buf, START_SEQUENCE);
    if (v->start_code != -1)
        decode_sequence_header(v);
    return v;
}

void orc_video_close(orc_video *v)
{
    if (!v)
        return;
    /* the three frames hold distinct allocations at all times (they rotate by value) */
    orc_frame_free(&v->cur);
    orc_frame_free(&v->fwd);
    orc_frame_free(&v->bwd);
    free(v);
}

int orc_video_has_header(orc_video *v)
{ /* video.go:130-147 */
    if (v->has_seq)
        return 1;
    if (v->start_code != START_SEQUENCE)
        v->start_code = bits_find_start_code(&v->buf, START_SEQUENCE);
    if (v->start_code == -1)
        return 0;
    return decode_sequence_header(v);
}

int orc_video_width(orc_video *v) { return orc_video_has_header(v) ? v->width : 0; }
int orc_video_height(orc_video *v) { return orc_video_has_header(v) ? v->height : 0; }
double orc_video_framerate(orc_video *v) { return orc_video_has_header(v) ? v->frame_rate : 0; }
void orc_video_set_no_delay(orc_video *v, int nd) { v->no_delay = nd; }
const orc_video_stats *orc_video_get_stats(const orc_video *v) { return &v->st; }

/* Buffer.Rewind (buffer.go:105-107 -> seek(0), :158-176) over the resident stream: the reader is back at byte 0, nothing is loaded,
 * hasEnded is cleared. */
static void bits_rewind(orc_bits *b)
{
    b->bit = 0;
    b->ended = 0;
}
/* Video.Rewind (video.go:195-201).  The three frames keep their bytes and the rotation its state: what the first pictures after a
 * rewind predict from — on a stream that does not open with an intra picture, or a damaged one — is what was decoded before it. */
void orc_video_rewind(orc_video *v)
{
    bits_rewind(&v->buf);
    v->time = 0;
    v->frames_decoded = 0;
    v->has_ref = 0;
    v->start_code = -1;
}
double orc_video_time(const orc_video *v) { return v->time; }             /* video.go:183 */
int orc_video_has_ended(const orc_video *v) { return v->buf.ended; }        /* video.go:203 */

const orc_frame *orc_video_decode(orc_video *v)
{ /* video.go:209-268 */
    if (!orc_video_has_header(v))
        return NULL;
    orc_frame *frame = NULL;
    for (;;) {
        if (v->start_code != START_PICTURE) {
            v->start_code = bits_find_start_code(&v->buf, START_PICTURE);
            if (v->start_code == -1) {
                if (v->has_ref && !v->no_delay && v->buf.ended &&
                    (v->picture_type == PIC_I || v->picture_type == PIC_P)) {
                    v->has_ref = 0;
                    frame = &v->bwd;
                    break;
                }
                return NULL;
            }
        }
        if (bits_has_start_code(&v->buf, START_PICTURE) == -1 && !v->buf.ended)
            return NULL;
        decode_picture(v);
        if (v->no_delay)
            frame = &v->bwd;
        else if (v->picture_type == PIC_B)
            frame = &v->cur;
        else if (v->has_ref)
            frame = &v->fwd;
        else
            v->has_ref = 1;
        if (frame)
            break;
    }
    frame->time = v->time;
    v->frames_decoded++;
    v->time = (double)v->frames_decoded / v->frame_rate;
    v->st.frames_returned++;
    return frame;
}

/* ====================================================================== audio */

/* Matrixing DCT constants: c_N[i] = 0.5 / cos((2i+1)*pi/(2N)).  The decimal
 * literals of audio.go:498-661 round to the same float32 values (checked). */
static const float k_c32[16] = {
    0.50060299823519630f, 0.50547095989754365f, 0.51544730992262455f, 0.53104259108978417f,
    0.55310389603444452f, 0.58293496820613389f, 0.62250412303566482f, 0.67480834145500568f,
    0.74453627100229858f, 0.83934964541552681f, 0.97256823786196078f, 1.16943993343288470f,
    1.48416461631416620f, 2.05778100995341100f, 3.40760841846871900f, 10.19000812354803300f};
static const float k_c16[8] = {
    0.50241928618815568f, 0.52249861493968885f, 0.56694403481635769f, 0.64682178335999008f,
    0.78815462345125020f, 1.06067768599034740f, 1.72244709823833420f, 5.10114861868915500f};
static const float k_c8[4] = {0.50979557910415918f, 0.60134488693504529f, 0.89997622313641557f, 2.56291544774150550f};
static const float k_c4[2] = {0.54119610014619701f, 1.30656296487637640f};
static const float k_c2[1] = {0.70710678118654746f};

static const float *dct_coef(int n)
{
    switch (n) {
    case 32: return k_c32;
    case 16: return k_c16;
    case 8:  return k_c8;
    case 4:  return k_c4;
    default: return k_c2;
    }
}

/* The butterfly network of audio.go:530-706 is the recursive even/odd split
 *   e[i] = x[i] + x[n-1-i],  o[i] = (x[i] - x[n-1-i]) * c_n[i]
 *   E = dct(e), O = dct(o), O[k] += O[k+1] (k ascending), X[2k] = E[k], X[2k+1] = O[k]
 * written out for n = 16 (twice), 8, 4, 2.  Same operations on the same
 * operands in the same association order => bit-identical float32 results. */
static void dct_rec(float *x, int n)
{
    if (n == 1)
        return;
    int h = n / 2;
    float e[16], o[16];
    const float *c = dct_coef(n);
    for (int i = 0; i < h; i++) {
        e[i] = x[i] + x[n - 1 - i];
        o[i] = (x[i] - x[n - 1 - i]) * c[i];
    }
    dct_rec(e, h);
    dct_rec(o, h);
    for (int k = 0; k + 1 < h; k++)
        o[k] += o[k + 1];
    for (int k = 0; k < h; k++) {
        x[2 * k] = e[k];
        x[2 * k + 1] = o[k];
    }
}

void orc_idct36(const int64_t s[32][3], int ss, float d[1024], int dp)
{
    float e[16], o[16], X[32];
    for (int i = 0; i < 16; i++) { /* audio.go:497-528: integer add/sub, then float32 */
        e[i] = (float)(s[i][ss] + s[31 - i][ss]);
        o[i] = (float)(s[i][ss] - s[31 - i][ss]) * k_c32[i];
    }
    dct_rec(e, 16);
    dct_rec(o, 16);
    for (int k = 0; k < 15; k++) /* audio.go:692-706 */
        o[k] += o[k + 1];
    for (int k = 0; k < 16; k++) {
        X[2 * k] = e[k];
        X[2 * k + 1] = o[k];
    }
    /* audio.go:708-771: mirrored / negated scatter into 64 ring entries */
    for (int k = 0; k <= 16; k++)
        d[dp + 48 - k] = -X[k];
    for (int k = 1; k <= 15; k++)
        d[dp + 48 + k] = -X[k];
    for (int k = 17; k <= 31; k++) {
        d[dp + 48 - k] = -X[k];
        d[dp + k - 16] = X[k];
    }
    d[dp + 0] = X[16];
    d[dp + 16] = 0.0f;
}

void orc_window_table(float d[1024])
{ /* audio.go:95-98 */
    for (int i = 0; i < 512; i++) {
        float w = (float)orc_synth_window_x2[i] * 0.5f; /* exact */
        d[i] = w;
        d[i + 512] = w;
    }
}

static inline float mac(float acc, float a, float b, int fma)
{
    return fma ? fmaf(a, b, acc) : acc + a * b; /* -ffp-contract=off keeps the second form unfused */
}

void orc_synth_window(float u[32], const float d[1024], const float v[1024], int vpos, int fma)
{ /* audio_noasm.go:8-38 */
    for (int i = 0; i < 32; i++)
        u[i] = 0;
    int di = 512 - (vpos >> 1);
    int vi = (vpos % 128) >> 1;
    while (vi < 1024) {
        for (int i = 0; i < 32; i++)
            u[i] = mac(u[i], d[di + i], v[vi + i], fma);
        vi += 128;
        di += 64;
    }
    di -= 512 - 32;
    vi = (128 - 32 + 1024) - vi;
    while (vi < 1024) {
        for (int i = 0; i < 32; i++)
            u[i] = mac(u[i], d[di + i], v[vi + i], fma);
        vi += 128;
        di += 64;
    }
}

void orc_synth_window_ref(float u[32], const float d[1024], const float v[1024], int vpos, int fma)
{ /* audio_test.go:9-31 — tap-major statement of the same sum */
    for (int i = 0; i < 32; i++) {
        float acc = 0;
        int v0 = (vpos % 128) >> 1, d0 = 512 - (vpos >> 1);
        for (int k = 0; k < 8; k++)
            acc = mac(acc, d[d0 + 64 * k + i], v[v0 + 128 * k + i], fma);
        for (int k = 0; k < 8; k++)
            acc = mac(acc, d[d0 + 32 + 64 * k + i], v[96 - v0 + 128 * k + i], fma);
        u[i] = acc;
    }
}

/* ISO 11172-3 Layer II tables, audio.go:798-973 */
static const uint16_t k_samplerate[4] = {44100, 48000, 32000, 0};
static const int16_t k_bitrate[14] = {32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384};
static const int k_sf_base[3] = {0x02000000, 0x01965FEA, 0x01428A30};
static const uint8_t k_q1[2][14] = {
    {0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2},
    {0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2}};
#define QT_A (27 | 64)
#define QT_B (30 | 64)
#define QT_C 8
#define QT_D 12
static const uint8_t k_q2[3][3] = {{QT_C, QT_C, QT_D}, {QT_A, QT_A, QT_A}, {QT_B, QT_A, QT_B}};
static const uint8_t k_q3[2][32] = {
    {0x44, 0x44, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34},
    {0x43, 0x43, 0x43, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x31, 0x31, 0x31, 0x31, 0x31,
     0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20}};
static const uint8_t k_q4[6][16] = {
    {0, 1, 2, 17},
    {0, 1, 2, 3, 4, 5, 6, 17},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 17},
    {0, 1, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17},
    {0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}};
typedef struct { uint16_t levels; uint8_t group, bits; } quant_spec;
static const quant_spec k_qtab[17] = {
    {3, 1, 5}, {5, 1, 7}, {7, 0, 3}, {9, 1, 10}, {15, 0, 4}, {31, 0, 5}, {63, 0, 6}, {127, 0, 7},
    {255, 0, 8}, {511, 0, 9}, {1023, 0, 10}, {2047, 0, 11}, {4095, 0, 12}, {8191, 0, 13},
    {16383, 0, 14}, {32767, 0, 15}, {65535, 0, 16}};

enum { MODE_STEREO = 0, MODE_JOINT = 1, MODE_DUAL = 2, MODE_MONO = 3 };

struct orc_audio {
    orc_bits buf;
    double time;
    int samples_decoded;
    int samplerate_index, bitrate_index, version, layer, mode, channels, bound, vpos;
    int next_frame_data_size, has_header;
    const quant_spec *alloc[2][32];
    uint8_t scfsi[2][32];
    int scale_factor[2][32][3];
    int64_t sample[2][32][3];
    float interleaved[2304];
    float d[1024], v[2][1024], u[32];
    int fma;
};

static int find_frame_sync(orc_bits *b)
{ /* buffer.go:326-339 */
    size_t i;
    for (i = b->bit >> 3; i + 1 < b->len; i++) {
        if (b->data[i] == 0xFF && (b->data[i + 1] & 0xFE) == 0xFC) {
            b->bit = ((i + 1) << 3) + 3;
            return 1;
        }
    }
    b->bit = (i + 1) << 3;
    return 0;
}

static int audio_decode_header(orc_audio *a)
{ /* audio.go:184-272 */
    orc_bits *b = &a->buf;
    if (!bits_has(b, 48))
        return 0;
    bits_skip_bytes(b, 0x00);
    int sync = (int)bits_read(b, 11);
    if (sync != 0x7ff && !find_frame_sync(b))
        return 0;
    a->version = (int)bits_read(b, 2);
    a->layer = (int)bits_read(b, 2);
    int has_crc = bits_read1(b) == 0;
    if (a->version != 0x3 || a->layer != 0x2)
        return 0;
    int bitrate_index = (int)bits_read(b, 4) - 1;
    if (bitrate_index > 13)
        return 0;
    int samplerate_index = (int)bits_read(b, 2);
    if (samplerate_index == 3)
        return 0;
    int padding = bits_read1(b);
    bits_skip(b, 1);
    int mode = (int)bits_read(b, 2);
    if (a->has_header && (a->bitrate_index != bitrate_index || a->samplerate_index != samplerate_index || a->mode != mode))
        return 0;
    if (bitrate_index < 0)
        return 0; /* "free format": Go indexes bitrate[-1] and panics; not on the fixtures */
    a->bitrate_index = bitrate_index;
    a->samplerate_index = samplerate_index;
    a->mode = mode;
    a->has_header = 1;
    if (mode == MODE_STEREO || mode == MODE_JOINT)
        a->channels = 2;
    else if (mode == MODE_MONO)
        a->channels = 1;
    if (mode == MODE_JOINT) {
        a->bound = ((int)bits_read(b, 2) + 1) << 2;
    } else {
        bits_skip(b, 2);
        a->bound = mode == MODE_MONO ? 0 : 32;
    }
    bits_skip(b, 4);
    if (has_crc)
        bits_skip(b, 16);
    int frame_size = (144000 * (int)k_bitrate[a->bitrate_index] / (int)k_samplerate[a->samplerate_index]) + padding;
    return frame_size - (has_crc ? 6 : 4);
}

static const quant_spec *read_allocation(orc_audio *a, int sb, int tab3)
{ /* audio.go:429-438 */
    int tab4 = k_q3[tab3][sb];
    int qtab = k_q4[tab4 & 15][bits_read(&a->buf, tab4 >> 4)];
    return qtab ? &k_qtab[qtab - 1] : NULL;
}

static void read_samples(orc_audio *a, int ch, int sb, int part)
{ /* audio.go:440-490 */
    const quant_spec *q = a->alloc[ch][sb];
    int64_t sf = a->scale_factor[ch][sb][part];
    int64_t *s = a->sample[ch][sb];
    if (!q) {
        s[0] = s[1] = s[2] = 0;
        return;
    }
    if (sf == 63) {
        sf = 0;
    } else {
        int shift = (int)(sf / 3);
        sf = (k_sf_base[sf % 3] + ((1 << shift) >> 1)) >> shift;
    }
    int64_t adj = q->levels;
    if (q->group) {
        int64_t val = bits_read(&a->buf, q->bits);
        s[0] = val % adj;
        val /= adj;
        s[1] = val % adj;
        s[2] = val / adj;
    } else {
        s[0] = bits_read(&a->buf, q->bits);
        s[1] = bits_read(&a->buf, q->bits);
        s[2] = bits_read(&a->buf, q->bits);
    }
    int64_t scale = 65536 / (adj + 1);
    adj = ((adj + 1) >> 1) - 1;
    for (int k = 0; k < 3; k++) {
        int64_t val = (adj - s[k]) * scale;
        s[k] = (val * (sf >> 12) + ((val * (sf & 4095) + 2048) >> 12)) >> 12;
    }
}

static void audio_decode_frame(orc_audio *a, int32_t *samples_out)
{ /* audio.go:274-427 */
    orc_bits *b = &a->buf;
    int tab1 = a->mode == MODE_MONO ? 0 : 1;
    int tab2 = k_q1[tab1][a->bitrate_index];
    int tab3 = k_q2[tab2][a->samplerate_index];
    int sblimit = tab3 & 63;
    tab3 >>= 6;
    if (a->bound > sblimit)
        a->bound = sblimit;

    for (int sb = 0; sb < a->bound; sb++) {
        a->alloc[0][sb] = read_allocation(a, sb, tab3);
        a->alloc[1][sb] = read_allocation(a, sb, tab3);
    }
    for (int sb = a->bound; sb < sblimit; sb++) {
        a->alloc[0][sb] = read_allocation(a, sb, tab3);
        a->alloc[1][sb] = a->alloc[0][sb];
    }
    int channels = a->mode == MODE_MONO ? 1 : 2;
    for (int sb = 0; sb < sblimit; sb++) {
        for (int ch = 0; ch < channels; ch++)
            if (a->alloc[ch][sb])
                a->scfsi[ch][sb] = (uint8_t)bits_read(b, 2);
        if (a->mode == MODE_MONO)
            a->scfsi[1][sb] = a->scfsi[0][sb];
    }
    for (int sb = 0; sb < sblimit; sb++) {
        for (int ch = 0; ch < channels; ch++) {
            if (!a->alloc[ch][sb])
                continue;
            int *sf = a->scale_factor[ch][sb];
            switch (a->scfsi[ch][sb]) {
            case 0:
                sf[0] = (int)bits_read(b, 6);
                sf[1] = (int)bits_read(b, 6);
                sf[2] = (int)bits_read(b, 6);
                break;
            case 1:
                sf[0] = sf[1] = (int)bits_read(b, 6);
                sf[2] = (int)bits_read(b, 6);
                break;
            case 2:
                sf[0] = sf[1] = sf[2] = (int)bits_read(b, 6);
                break;
            case 3:
                sf[0] = (int)bits_read(b, 6);
                sf[1] = sf[2] = (int)bits_read(b, 6);
                break;
            }
        }
        if (a->mode == MODE_MONO)
            memcpy(a->scale_factor[1][sb], a->scale_factor[0][sb], sizeof(a->scale_factor[0][sb]));
    }

    int out_pos = 0, t = 0;
    for (int part = 0; part < 3; part++) {
        for (int granule = 0; granule < 4; granule++) {
            for (int sb = 0; sb < a->bound; sb++) {
                read_samples(a, 0, sb, part);
                read_samples(a, 1, sb, part);
            }
            for (int sb = a->bound; sb < sblimit; sb++) {
                read_samples(a, 0, sb, part);
                memcpy(a->sample[1][sb], a->sample[0][sb], sizeof(a->sample[0][sb]));
            }
            for (int sb = sblimit; sb < 32; sb++) {
                memset(a->sample[0][sb], 0, sizeof(a->sample[0][sb]));
                memset(a->sample[1][sb], 0, sizeof(a->sample[1][sb]));
            }
            for (int p = 0; p < 3; p++, t++) {
                a->vpos = (a->vpos - 64) & 1023;
                for (int ch = 0; ch < 2; ch++) {
                    if (samples_out)
                        for (int sb = 0; sb < 32; sb++)
                            samples_out[(ch * 36 + t) * 32 + sb] = (int32_t)a->sample[ch][sb][p];
                    orc_idct36((const int64_t(*)[3])a->sample[ch], p, a->v[ch], a->vpos);
                    orc_synth_window(a->u, a->d, a->v[ch], a->vpos, a->fma);
                    for (int j = 0; j < 32; j++) /* audio.go:388-391, AudioF32N */
                        a->interleaved[((out_pos + j) << 1) + ch] = a->u[j] / -1090519040.0f;
                }
                out_pos += 32;
            }
        }
    }
    bits_align(b);
}

orc_audio *orc_audio_open(const uint8_t *data, size_t len, int fma)
{ /* audio.go:83-104 */
    orc_audio *a = (orc_audio *)calloc(1, sizeof(*a));
    if (!a)
        return NULL;
    a->buf.data = data;
    a->buf.len = len;
    a->samplerate_index = 3;
    a->fma = fma;
    orc_window_table(a->d);
    a->next_frame_data_size = audio_decode_header(a);
    return a;
}

void orc_audio_close(orc_audio *a) { free(a); }

int orc_audio_samplerate(orc_audio *a)
{ /* audio.go:112-129 */
    if (!a->has_header)
        a->next_frame_data_size = audio_decode_header(a);
    return a->has_header ? k_samplerate[a->samplerate_index] : 0;
}

int orc_audio_channels(orc_audio *a) { return a->channels; }

/* Audio.Rewind (audio.go:149-154): the V ring and vPos are NOT cleared — the first frames after a rewind are synthesised on top
 * of what the frames before it left behind. */
void orc_audio_rewind(orc_audio *a)
{
    a->buf.bit = 0;
    a->buf.ended = 0;
    a->time = 0;
    a->samples_decoded = 0;
    a->next_frame_data_size = 0;
}
double orc_audio_time(const orc_audio *a) { return a->time; }              /* audio.go:137 */
int orc_audio_has_ended(const orc_audio *a) { return a->buf.ended; }        /* audio.go:157 */

const float *orc_audio_decode(orc_audio *a, int32_t *samples_out)
{ /* audio.go:163-182 */
    if (a->next_frame_data_size == 0)
        a->next_frame_data_size = audio_decode_header(a);
    if (a->next_frame_data_size == 0 || !bits_has(&a->buf, (size_t)a->next_frame_data_size << 3))
        return NULL;
    audio_decode_frame(a, samples_out);
    a->next_frame_data_size = 0;
    a->samples_decoded += 1152;
    a->time = (double)a->samples_decoded / (double)k_samplerate[a->samplerate_index];
    return a->interleaved;
}

void orc_audio_get_state(const orc_audio *a, float v[2][1024], int *vpos)
{
    memcpy(v, a->v, sizeof(a->v));
    *vpos = a->vpos;
}

/* ================================================================ PS payloads */

static void skip_time(orc_bits *b) /* demux.go:518-529 decodeTime, value unused */
{
    bits_read(b, 3); bits_skip(b, 1); bits_read(b, 15); bits_skip(b, 1); bits_read(b, 15); bits_skip(b, 1);
}

uint8_t *orc_ps_extract(const uint8_t *data, size_t len, int type, size_t *out_len, int *n_packets)
{ /* demux.go:473-584: Decode -> decodePacket -> packet, headers not required */
    orc_bits b = {data, len, 0, 0};
    uint8_t *out = (uint8_t *)malloc(len ? len : 1);
    size_t n = 0;
    int packets = 0;
    for (;;) {
        int code = bits_next_start_code(&b);
        if (code == -1)
            break;
        if (!(code == 0xE0 || code == 0xBD || (code >= 0xC0 && code <= 0xC3)))
            continue;
        if (!bits_has(&b, 16 << 3))
            break;
        int64_t length = bits_read(&b, 16);
        length -= bits_skip_bytes(&b, 0xff);
        if (bits_read(&b, 2) == 0x01) {
            bits_skip(&b, 16);
            length -= 2;
        }
        int marker = (int)bits_read(&b, 2);
        if (marker == 0x03) {
            skip_time(&b);
            bits_skip(&b, 40);
            length -= 10;
        } else if (marker == 0x02) {
            skip_time(&b);
            length -= 5;
        } else if (marker == 0x00) {
            bits_skip(&b, 4);
            length -= 1;
        } else {
            continue; /* invalid */
        }
        if (length < 0 || !bits_has(&b, (size_t)length << 3))
            break;
        if (code == type) {
            memcpy(out + n, data + (b.bit >> 3), (size_t)length);
            n += (size_t)length;
            packets++;
        }
        b.bit += (size_t)length << 3;
    }
    *out_len = n;
    if (n_packets)
        *n_packets = packets;
    return out;
}
