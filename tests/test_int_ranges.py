"""Why int32 and 24-bit multiplies are exact on the device.

The reference computes the IDCT in 64-bit Go ints (video.go:101).  The kernels use int32 adds and
v_mul_i32_i24 (24-bit signed operands).  Every IDCT intermediate is a linear form of the 64 inputs plus
bounded rounding terms; with |block[j]| <= 2048 * premultiplier[j] for every AC entry (clip at
video.go:737-741, premultiply :744) and |block[0]| <= 2^30 (the emitter saturates an intra DC there,
which is exact because such a DC saturates the pixel), the worst case over all sign patterns is the L1
norm.  This test propagates those bounds through the exact operation sequence of idct8 (video_lane.h)
for both passes and asserts: multiplicands < 2^23, everything < 2^31."""
import numpy as np

from mpeg_amd.synth import PREMULT


class Form:
    """linear form over the 64 inputs + an absolute slack for accumulated rounding"""

    def __init__(self, c=None, slack=0.0):
        self.c = np.zeros(64) if c is None else c
        self.slack = slack

    def __add__(self, o):
        return Form(self.c + o.c, self.slack + o.slack)

    def __sub__(self, o):
        return Form(self.c - o.c, self.slack + o.slack)

    def __neg__(self):
        return Form(-self.c, self.slack)

    def scale_round_shift8(self, k):  # (x*k + 128) >> 8
        return Form(self.c * (k / 256.0), self.slack * (abs(k) / 256.0) + 1.0)

    def bound(self, lim):
        return float(np.abs(self.c) @ lim + self.slack)


def idct8_forms(v, final_shift, lim, record):
    def mulcheck(x, k):
        record("multiplicand", x.bound(lim))
        record("product", x.bound(lim) * abs(k))
        return x
    b1 = v[4]
    b3 = v[2] + v[6]
    b4 = v[5] - v[3]
    tmp1 = v[1] + v[7]
    tmp2 = v[3] + v[5]
    b6 = v[1] - v[7]
    b7 = tmp1 + tmp2
    m0 = Form(v[0].c, v[0].slack + (128.0 if final_shift else 0.0))   # the row pass adds the final rounding's 128 to m0 (idct8)
    mulcheck(b6, 473), mulcheck(b4, 196)
    s1 = Form(b6.c * 473 - b4.c * 196, b6.slack * 473 + b4.slack * 196 + 128)
    record("sum", s1.bound(lim))
    x4 = Form(s1.c / 256.0, s1.slack / 256.0 + 1) - b7
    d12 = tmp1 - tmp2
    mulcheck(d12, 362)
    x0 = x4 - d12.scale_round_shift8(362)
    x1 = m0 - b1
    d26 = v[2] - v[6]
    mulcheck(d26, 362)
    x2 = d26.scale_round_shift8(362) - b3
    x3 = m0 + b1
    y3, y4, y5, y6 = x1 + x2, x3 + b3, x1 - x2, x3 - b3
    s2 = Form(b4.c * 473 + b6.c * 196, b4.slack * 473 + b6.slack * 196 + 128)
    record("sum", s2.bound(lim))
    y7 = -x0 - Form(s2.c / 256.0, s2.slack / 256.0 + 1)
    outs = [b7 + y4, x4 + y3, y5 - x0, y6 - y7, y6 + y7, x0 + y5, y3 - x4, y4 - b7]
    for t in (b3, b4, tmp1, tmp2, b6, b7, x4, x0, x1, x2, x3, y3, y4, y5, y6, y7, *outs):
        record("sum", t.bound(lim))
    if final_shift:
        outs = [Form(o.c / 256.0, o.slack / 256.0 + 1.0) for o in outs]
    return outs


def test_idct_fits_int32_and_mul24():
    lim = 2048.0 * PREMULT.astype(float)
    lim[0] = float(1 << 30)                      # saturated DC (never multiplied)
    worst = {"multiplicand": 0.0, "product": 0.0, "sum": 0.0}

    def record(kind, b):
        worst[kind] = max(worst[kind], b)

    block = [[Form(np.eye(64)[r * 8 + c]) for c in range(8)] for r in range(8)]
    # column pass (no final shift): column c takes rows 0..7
    for c in range(8):
        outs = idct8_forms([block[r][c] for r in range(8)], False, lim, record)
        for r in range(8):
            block[r][c] = outs[r]
    # row pass
    for r in range(8):
        idct8_forms(block[r], True, lim, record)
    assert worst["multiplicand"] < 2 ** 23, worst   # v_mul_i32_i24 operand range
    assert worst["product"] < 2 ** 31, worst
    assert worst["sum"] < 2 ** 31, worst
    # the survey's figure for the AC-only worst case (DC excluded) is 0.30 * 2^31; with the saturated DC
    # term the sums stay below 2^31 as well
    print(worst)


def test_dequant_fits_mul24():
    # |2q + sign| <= 513 (q in [-256, 255]); quantiser_scale * matrix <= 31 * 255; |level| <= 2048, premult <= 62
    assert 513 * 31 * 255 < 2 ** 23 * 1 and 513 < 2 ** 23 and 31 * 255 < 2 ** 23
    assert 513 * 31 * 255 < 2 ** 31 and 2048 * 62 < 2 ** 23
