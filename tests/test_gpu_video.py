"""-m gpu: the HIP reconstruction path, called through the C ABI (libmpeghip.so),
against the CPU oracle on the same seeded descriptor batches.  Bit-exact."""
import numpy as np
import pytest

from mpeg_amd import abi, desc, synth
from parity import assert_planes_equal, run_and_compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 30, "typical", 0.0, False),    # BASELINE config 2: SIF, one GOP of 30 pictures
    (352, 240, 8, "typical", 0.15, False),    # with int32 snapshot (COEF_RAW) blocks
    (352, 240, 6, "dense", 0.0, False),       # every block full, odd vectors
    (160, 120, 10, "typical", 0.05, True),    # fused Frame.RGBA
    (176, 144, 5, "typical", 0.0, True),
    (24, 40, 4, "typical", 0.0, True),
    (50, 35, 3, "typical", 0.0, True),        # odd height: the last RGBA row has no partner
    (1920, 1080, 4, "typical", 0.0, True),    # BASELINE config 3: 1080p, fused IDCT+MC+RGBA
    (1920, 1080, 2, "dense", 0.0, False),
    (1920, 1080, 2, "dense", 0.0, True),      # BASELINE config 3's worst case: dense + fused RGBA at 1080p
])
@pytest.mark.parametrize("policy", [0, 1], ids=["auto", "pinned_int16"])
def test_reconstruction_bit_exact(oracle, hip_ctx, w, h, n, profile, raw, rgba, policy):
    """(policy 0: the library's choice — recon_wide_kernel for these one-picture launches; 1: recon_kernel's int16-tile instance)"""
    seq = synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba)
    ref, dut = oracle.OracleStore(w, h, threads=4), abi.VideoStore(hip_ctx, w, h)
    dut.set_tile_policy(policy)
    try:
        run_and_compare(ref, dut, seq, check_rgba=rgba)
    finally:
        dut.close()
        ref.close()


@pytest.mark.parametrize("w,h", [(352, 240), (160, 120)])                       # mb_w = 22, 10: a row's last chunk carries on into the next row
@pytest.mark.parametrize("rgba", [False, True], ids=["planes", "rgba_fused"])
@pytest.mark.parametrize("policy", [0, 1, 2], ids=["auto_wide", "pinned_dpp", "pinned_lds"])
def test_runs_across_row_ends(oracle, hip_ctx, w, h, rgba, policy):
    """A chunk whose 4 macroblocks are consecutive in raster order across a row end is a run for the plane stores (round 6:
    rc_run_follows) and converts to RGBA macroblock by macroblock (rc_run_in_one_row): all three kernels (recon_wide_kernel, both
    recon_kernel instances), host-packed units and device-packed sparse pictures, against the oracle."""
    seq = synth.generate_sequence(w, h, 7, profile="typical", rgba=rgba, seed=0x77)
    ref, units, packed = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h), abi.VideoStore(hip_ctx, w, h)
    units.set_tile_policy(policy)
    packed.set_tile_policy(policy)
    try:
        for i, s in enumerate(seq):
            ref.submit(s.pics, s.mbs, s.coefs)
            units.submit(s.pics, s.mbs, s.coefs)
            mbs, words = desc.to_sparse(s.mbs, s.coefs)
            packed.submit_staged_device([(s.pics[0], mbs, words)])
            for slot in range(3):
                want = ref.read_planes(0, slot)
                assert_planes_equal(want, units.read_planes(0, slot), "units, picture %d slot %d" % (i, slot))
                assert_planes_equal(want, packed.read_planes(0, slot), "device-packed, picture %d slot %d" % (i, slot))
            if rgba:
                want = ref.read_rgba(0, s.cur)
                assert np.array_equal(want, units.read_rgba(0, s.cur)) and np.array_equal(want, packed.read_rgba(0, s.cur)), i
    finally:
        units.close()
        packed.close()
        ref.close()


def test_custom_quant_matrices(oracle, hip_ctx):
    rng = np.random.default_rng(5)
    iq, nq = rng.integers(1, 256, 64), rng.integers(1, 256, 64)
    ref, dut = oracle.OracleStore(64, 48), abi.VideoStore(hip_ctx, 64, 48)
    ref.set_quant(0, iq, nq)
    dut.set_quant(0, iq, nq)
    run_and_compare(ref, dut, synth.generate_sequence(64, 48, 6, seed=11))
    dut.close()


@pytest.mark.parametrize("case", ["default", "custom_intra_default_non_intra", "first_entry_off", "last_entry_off"])
def test_dense_units_under_the_default_and_nearly_default_non_intra_matrix(oracle, hip_ctx, case):
    """the short dequantisation of dense non-intra units (rc_dense_cols<true>) runs only when every entry of the stream's
    non-intra matrix is 16; both kernel instances"""
    rng = np.random.default_rng(21)
    w, h = 96, 64
    seq = synth.generate_sequence(w, h, 4, seed=77, profile="dense")
    for sub in seq:
        sub.mbs["qscale"] = rng.integers(1, 32, size=len(sub.mbs))
    iq, nq = rng.integers(1, 256, 64), np.full(64, 16)
    if case == "first_entry_off":
        nq[0] = 17
    elif case == "last_entry_off":
        nq[63] = 15
    for policy in (1, 2):
        ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)
        dut.set_tile_policy(policy)
        if case != "default":
            ref.set_quant(0, iq, nq)
            dut.set_quant(0, iq, nq)
        run_and_compare(ref, dut, seq)
        dut.close()


@pytest.mark.parametrize("w,h", [(100, 60), (37, 23), (16, 1), (3, 16)])  # odd sizes: partial quads, a last row without a partner
def test_standalone_rgba(oracle, hip_ctx, w, h):
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h, 2)
    rng = np.random.default_rng(3)
    g = desc.geometry(w, h)
    y, cb, cr = (rng.integers(0, 256, n, dtype=np.uint8) for n in (g["luma_bytes"], g["chroma_bytes"], g["chroma_bytes"]))
    ref.write_planes(0, 1, y, cb, cr)
    dut.write_planes(1, 1, y, cb, cr)
    dut.rgba_convert(1)
    assert np.array_equal(ref.read_rgba(0, 1), dut.read_rgba(1, 1))
    # untouched stream: RGBA of the all-zero frame
    zero = oracle.OracleStore(w, h)
    assert np.array_equal(zero.read_rgba(0, 0), dut.read_rgba(0, 0))
    dut.close()


def test_rgba_of_every_possible_pixel(oracle, hip_ctx):
    """All 2^24 (Y, Cb, Cr) triples through the device's conversion, against the oracle's: 64 frames of 512x512,
    chroma sample (cx, cy) = (Cb, Cr) = (cx, cy), its four luma pixels in frame f = 4f .. 4f+3."""
    w = h = 512
    n = 64
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h, n)
    g = desc.geometry(w, h)
    cb = np.tile(np.arange(256, dtype=np.uint8), 256)
    cr = np.repeat(np.arange(256, dtype=np.uint8), 256)
    quad = (np.arange(h)[:, None] & 1) * 2 + (np.arange(w)[None, :] & 1)
    for f in range(n):
        y = (4 * f + quad).astype(np.uint8).reshape(-1)
        dut.write_planes(f, 1, y, cb, cr)
    dut.rgba_convert(1)
    for f in (0, 1, 17, 31, 32, 62, 63):    # the oracle converts a frame in ~10 ms; 7 frames sample every luma range
        y = (4 * f + quad).astype(np.uint8).reshape(-1)
        ref.write_planes(0, 1, y, cb, cr)
        assert np.array_equal(ref.read_rgba(0, 1), dut.read_rgba(f, 1)), f
    # and all 64 frames against the closed form (clamp((y*65793 + chroma term) >> 16, 0, 255)), vectorised
    cbi, cri = cb.astype(np.int64).reshape(256, 256) - 128, cr.astype(np.int64).reshape(256, 256) - 128
    up = lambda a: np.repeat(np.repeat(a, 2, axis=0), 2, axis=1)
    terms = [up(91881 * cri), up(-22554 * cbi - 46802 * cri), up(116130 * cbi)]
    for f in range(n):
        yy = (4 * f + quad).astype(np.int64) * 0x10101
        want = np.stack([np.clip((yy + t) >> 16, 0, 255) for t in terms] + [np.full((h, w), 255)], axis=-1).astype(np.uint8)
        assert np.array_equal(want, dut.read_rgba(f, 1)), f
    dut.close()


def test_edge_cases(oracle, hip_ctx):
    """Empty submits, a single macroblock, pictures that touch only part of a frame."""
    w, h = 64, 64
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)
    dut.submit(np.zeros(0, desc.PIC_DTYPE), np.zeros(0, desc.MB_DTYPE), np.zeros(0, np.uint8))
    seq = synth.generate_sequence(w, h, 3, seed=7)
    for s in seq:
        keep = np.arange(len(s.mbs))[::3]            # ragged: every third macroblock only
        mbs = s.mbs[keep].copy()
        pics = s.pics.copy()
        pics["mb_count"] = len(mbs)
        ref.submit(pics, mbs, s.coefs)
        dut.submit(pics, mbs, s.coefs)
        for slot in range(3):
            assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(0, slot), "ragged slot %d" % slot)
    one = seq[0]
    pics = one.pics.copy()
    pics["mb_count"] = 1
    ref.submit(pics, one.mbs[:1], one.coefs)
    dut.submit(pics, one.mbs[:1], one.coefs)
    assert_planes_equal(ref.read_planes(0, one.cur), dut.read_planes(0, one.cur), "single macroblock")
    dut.close()


def test_out_of_range_vector_is_rejected(hip_ctx):
    """The reference panics when a prediction reads before a plane / past the buffer; the ABI refuses the submit."""
    dut = abi.VideoStore(hip_ctx, 64, 64)
    pics = np.zeros(1, desc.PIC_DTYPE)
    pics["cur"], pics["fwd"], pics["bwd"], pics["mb_count"] = 0, 1, 2, 1
    mbs = np.zeros(1, desc.MB_DTYPE)
    mbs["flags"], mbs["mv_x"], mbs["mv_y"] = desc.MB_REF_FWD, -2, -2   # macroblock (0,0) reading above the plane
    with pytest.raises(abi.MpegHipError) as ei:
        dut.submit(pics, mbs, np.zeros(0, np.uint8))
    assert ei.value.code == abi.ERR_RANGE
    mbs["mv_x"], mbs["mv_y"], mbs["mb_x"], mbs["mb_y"] = 0, 200, 3, 3  # far below the pad
    with pytest.raises(abi.MpegHipError):
        dut.submit(pics, mbs, np.zeros(0, np.uint8))
    dut.close()


def test_malformed_descriptors_are_refused_and_nothing_is_launched(oracle, hip_ctx):
    """Every field a caller can get wrong: the submit returns an error code (the reference would panic on
    an out-of-bounds slice), the device is not touched, and the store keeps working afterwards."""
    w, h = 64, 48
    seq = synth.generate_sequence(w, h, 3, seed=21)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)
    run_and_compare(ref, dut, seq[:1])
    good = seq[1]
    before = [dut.read_planes(0, s) for s in range(3)]

    def corrupt(field, value, which="mbs", index=0):
        pics, mbs, coefs = good.pics.copy(), good.mbs.copy(), good.coefs.copy()
        (mbs if which == "mbs" else pics)[field][index] = value
        return pics, mbs, coefs

    coded = int(np.nonzero(good.mbs["cbp"])[0][0])
    cases = [
        corrupt("mb_x", 4), corrupt("mb_y", 3), corrupt("pic", 1), corrupt("cbp", 0x40, index=coded),
        corrupt("qscale", 0, index=coded), corrupt("qscale", 32, index=coded),
        corrupt("coef_off", 1 << 20, index=coded), corrupt("flags", desc.MB_REF_FWD | desc.MB_REF_BWD),
        corrupt("flags", desc.MB_INTRA | desc.MB_REF_FWD), corrupt("flags", 0),
        corrupt("stream", 1, "pics"), corrupt("cur", 3, "pics"), corrupt("fwd", 7, "pics"), corrupt("bwd", 200, "pics"),
        corrupt("mb_count", len(good.mbs) + 1, "pics"), corrupt("mb_first", 5, "pics"),
        corrupt("flags", 0x40, "pics"), corrupt("flags", 0x04, "pics"),   # undefined picture flag bits (they must not pick the coefficient form)
    ]
    # every macroblock names the first units of a buffer that holds only as many as the largest macroblock needs: the packed
    # form's buffers are sized from the coefficient buffer, blocks may not share units beyond it
    pics, mbs, coefs = good.pics.copy(), good.mbs.copy(), good.coefs.copy()
    per_mb = np.array([bin(int(c)).count("1") for c in mbs["cbp"]]) * np.where(mbs["flags"] & desc.MB_COEF_RAW, 2, 1)
    assert (per_mb > 0).sum() >= 2
    mbs["coef_off"] = 0
    cases.append((pics, mbs, coefs[:int(per_mb.max()) * desc.COEF_UNIT]))
    for pics, mbs, coefs in cases:
        with pytest.raises(abi.MpegHipError) as ei:
            dut.submit(pics, mbs, coefs)
        assert ei.value.code in (abi.ERR_INVALID, abi.ERR_RANGE)
    with pytest.raises(abi.MpegHipError):  # coefficient buffer not a whole number of 128-byte units
        dut.submit(good.pics, good.mbs, good.coefs[:-1])
    for s in range(3):
        assert_planes_equal(before[s], dut.read_planes(0, s), "slot %d after refused submits" % s)
    run_and_compare(ref, dut, seq[1:])  # and the stream continues bit-exactly
    dut.close()


def test_fused_rgba_keeps_the_image_in_step_with_the_planes(oracle, hip_ctx):
    """MPEGHIP_PIC_RGBA converts inside the reconstruction kernel only the macroblocks a picture writes; the
    image must still equal Frame.RGBA() of the whole frame (video.go:31-36) when pictures are partial, when
    some pictures are not flagged, and after write_planes — the cases where the whole-frame pass is owed."""
    w, h = 96, 64
    flagged = synth.generate_sequence(w, h, 6, seed=33, rgba=True)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)

    def submit(s, n_mbs=None, flag=True):
        pics, mbs = s.pics.copy(), s.mbs.copy()
        if n_mbs is not None:   # a picture that covers only its first n_mbs macroblocks (coefficients stay contiguous)
            mbs = mbs[:n_mbs]
            pics["mb_count"] = n_mbs
        pics["flags"] = desc.PIC_RGBA if flag else 0
        ref.submit(pics, mbs, s.coefs)
        dut.submit(pics, mbs, s.coefs)
        for slot in range(3):
            assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(0, slot), "slot %d" % slot)
        return int(pics["cur"][0])

    def image_matches(slot):
        return np.array_equal(ref.read_rgba(0, slot), dut.read_rgba(0, slot))

    cur = submit(flagged[0])                       # full + flagged: fused conversion alone
    assert image_matches(cur)
    cur = submit(flagged[1], n_mbs=7)              # partial + flagged over a slot whose image was never converted
    assert image_matches(cur)
    cur = submit(flagged[2], flag=False)           # not flagged: the image of this slot goes stale ...
    cur2 = submit(flagged[3], n_mbs=5)             # ... (another slot meanwhile)
    assert image_matches(cur2)
    for s in flagged[4:]:
        cur = submit(s, n_mbs=11)                  # partial pictures keep rotating over stale and fresh slots
        assert image_matches(cur)
    rng = np.random.default_rng(2)
    g = desc.geometry(w, h)
    y, cb, cr = (rng.integers(0, 256, n, dtype=np.uint8) for n in (g["luma_bytes"], g["chroma_bytes"], g["chroma_bytes"]))
    ref.write_planes(0, cur, y, cb, cr)
    dut.write_planes(0, cur, y, cb, cr)            # planes replaced behind the image's back
    s = flagged[2]
    pics = s.pics.copy()
    pics["cur"], pics["flags"], pics["mb_count"] = cur, desc.PIC_RGBA, 3
    ref.submit(pics, s.mbs[:3], s.coefs)
    dut.submit(pics, s.mbs[:3], s.coefs)
    assert image_matches(cur)
    dut.close()


def test_overread_into_next_plane_and_pad(oracle, hip_ctx):
    """Legal reads past a plane end: bottom-row macroblocks with downward half-pel vectors
    read the first rows of the next plane / the zero pad, exactly like the reference's shared buffer."""
    w, h = 48, 32
    g = desc.geometry(w, h)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)
    rng = np.random.default_rng(9)
    y, cb, cr = (rng.integers(0, 256, n, dtype=np.uint8) for n in (g["luma_bytes"], g["chroma_bytes"], g["chroma_bytes"]))
    for st in (ref, dut):
        st.write_planes(0, 1, y, cb, cr)
    pics = np.zeros(1, desc.PIC_DTYPE)
    pics["cur"], pics["fwd"], pics["bwd"] = 0, 1, 2
    cases = [(mx, my) for mx in (-3, 0, 1, 5) for my in (1, 3, 9, 31)]
    for mx, my in cases:
        mbs = np.zeros(g["mb_w"], desc.MB_DTYPE)
        mbs["mb_x"], mbs["mb_y"] = np.arange(g["mb_w"]), g["mb_h"] - 1
        mbs["flags"], mbs["mv_x"], mbs["mv_y"] = desc.MB_REF_FWD, mx, my
        ok = synth.mv_in_range(g, mbs["mb_x"], mbs["mb_y"], mbs["mv_x"], mbs["mv_y"])
        mbs = mbs[ok]
        pics["mb_count"] = len(mbs)
        ref.submit(pics, mbs, np.zeros(0, np.uint8))
        dut.submit(pics, mbs, np.zeros(0, np.uint8))
        assert_planes_equal(ref.read_planes(0, 0), dut.read_planes(0, 0), "mv (%d,%d)" % (mx, my))
    dut.close()


@pytest.mark.parametrize("n_streams", [24, 25, 64, 86, 87, 128])
def test_many_streams_replicated_batch(oracle, hip_ctx, n_streams):
    """BASELINE config 5 in miniature: independent streams, one descriptor set each, no cross-talk.
    Property: identical inputs => identical FNV-1a-64 per stream, equal to the oracle's.
    (A SIF picture is 83 chunks: on a 256-CU part 24 streams = 1 992 chunks are the last launch recon_wide_kernel takes — four
    waves per chunk still fit the 8 192 wave slots —, 25 = 2 075 the first one back on recon_kernel; 86 streams = 7 138 chunks
    still run one chunk per wave, 87 = 7 221 take the two-chunk instances — launch_batch; a last wave with a single chunk at odd
    totals.)"""
    w, h = 352, 240
    seq = synth.generate_sequence(w, h, 6, seed=21)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h, n_streams)
    for s in seq:
        ref.submit(s.pics, s.mbs, s.coefs)
        b = dut.upload(s.pics, s.mbs, s.coefs, replicate=n_streams)
        assert b.n_mbs == len(s.mbs) * n_streams
        assert b.alg_bytes == synth.alg_bytes(desc.geometry(w, h), s) * n_streams
        b.run()
        b.free()
        want = 0
        for slot in range(3):
            hashes = dut.hash_slots(slot)
            want = oracle.FNV_OFFSET
            for p in ref.read_planes(0, slot):
                want = oracle.fnv1a64(p, want)
            assert (hashes == np.uint64(want)).all(), "slot %d" % slot
    dut.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
def test_staged_submit_equals_plain_submit(oracle, hip_ctx, threads):
    """mpeghip_video_stage_begin / _put / _commit: one submit assembled picture by picture (puts from several
    host threads, each picture with its own coefficient array) reconstructs exactly what one plain submit of
    the merged arrays does; pictures of odd macroblock counts make chunks straddle picture boundaries."""
    w, h = 96, 80   # 6 x 5 = 30 macroblocks per picture: not a multiple of the 4-macroblock chunk
    n = 5
    # stream 3: every block full (its units travel as they are, not as sparse entries: video_wire_lane.h)
    seqs = [synth.generate_sequence(w, h, 4, seed=300 + i, rgba=(i == 1), profile="dense" if i == 3 else "typical") for i in range(n)]
    ref, dut = oracle.OracleStore(w, h, n), abi.VideoStore(hip_ctx, w, h, n)
    for step in range(4):
        pictures = []
        for sidx, seq in enumerate(seqs):
            s = seq[step]
            p = s.pics.copy()
            p["stream"] = sidx
            pictures.append((p[0], s.mbs, s.coefs))
            q = p.copy()
            q["mb_first"] = 0
            ref.submit(q, s.mbs, s.coefs)
        assert dut.submit_staged(pictures, threads=threads) == [0] * n
        for sidx in range(n):
            for slot in range(3):
                assert_planes_equal(ref.read_planes(sidx, slot), dut.read_planes(sidx, slot), "stream %d slot %d" % (sidx, slot))
        cur = int(seqs[1][step].pics[0]["cur"])
        assert np.array_equal(np.asarray(dut.read_rgba(1, cur)).reshape(-1), ref.read_rgba(1, cur).reshape(-1))
    dut.close()


@pytest.mark.gpu
def test_staged_submit_of_ragged_pictures(oracle, hip_ctx):
    """Pictures of different sizes in one staged submit: a full picture, an empty one (no macroblocks), one with a
    few macroblocks and no coefficients, one whose macroblock count is not a multiple of the 4-macroblock chunk —
    chunks straddle picture boundaries in every way; int32 snapshot blocks (two units each) travel too."""
    w, h = 96, 80
    rng = np.random.default_rng(5)
    full = [synth.generate_sequence(w, h, 3, seed=40 + i, raw_fraction=0.2 if i == 2 else 0.0) for i in range(4)]
    ref, dut = oracle.OracleStore(w, h, 4), abi.VideoStore(hip_ctx, w, h, 4)
    for step in range(3):
        pictures = []
        for sidx in range(4):
            s = full[sidx][step]
            p = s.pics.copy()
            p["stream"] = sidx
            mbs, coefs = s.mbs, s.coefs
            if sidx == 1:                      # nothing at all
                mbs, coefs = mbs[:0], coefs[:0]
            elif sidx == 3:                    # the first 7 macroblocks only, prediction / intra as they are
                mbs = mbs[:7].copy()
                units = int(mbs["coef_off"][-1]) + bin(int(mbs["cbp"][-1])).count("1")
                coefs = coefs[:units * desc.COEF_UNIT]
            pictures.append((p[0], mbs, coefs))
            q = p.copy()
            q["mb_first"], q["mb_count"] = 0, len(mbs)
            if len(mbs):
                ref.submit(q, mbs, coefs)
        assert dut.submit_staged(pictures, threads=2) == [0] * 4
        for sidx in range(4):
            for slot in range(3):
                assert_planes_equal(ref.read_planes(sidx, slot), dut.read_planes(sidx, slot), "step %d stream %d slot %d" % (step, sidx, slot))
    dut.close()


@pytest.mark.gpu
def test_staged_submit_refuses_a_bad_picture(hip_ctx):
    """A put that fails validation makes the commit fail with that error and launch nothing; the stage is over
    and the store accepts the next submit."""
    w, h = 96, 80
    s = synth.generate_sequence(w, h, 2, seed=7)[0]
    dut = abi.VideoStore(hip_ctx, w, h, 2)
    good = s.pics.copy()
    bad_mbs = s.mbs.copy()
    bad_mbs["mb_x"][3] = 99
    p1 = good.copy()
    p1["stream"] = 1
    before = [dut.read_planes(st, int(good[0]["cur"])) for st in range(2)]
    with pytest.raises(abi.MpegHipError, match="position"):
        dut.submit_staged([(good[0], s.mbs, s.coefs), (p1[0], bad_mbs, s.coefs)])
    for st in range(2):
        assert_planes_equal(before[st], dut.read_planes(st, int(good[0]["cur"])), "stream %d untouched" % st)
    dut.submit_staged([(good[0], s.mbs, s.coefs), (p1[0], s.mbs, s.coefs)])
    a, b = dut.read_planes(0, int(good[0]["cur"])), dut.read_planes(1, int(good[0]["cur"]))
    assert_planes_equal(a, b, "both streams decoded the same picture")
    dut.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_random_geometries_plain_and_staged(oracle, hip_ctx, seed):
    """Seeded sweep over picture sizes (also not multiples of 16, also one macroblock wide or high), stream
    counts, macroblock mixes, snapshot blocks and fused RGBA: every picture goes to one store by plain submit and to
    another by staged submit; both must equal the oracle after every picture, planes and RGBA images."""
    rng = np.random.default_rng(1000 + seed)
    w = int(rng.choice([16, 17, 48, 100, 176, 250, 352]))
    h = int(rng.choice([16, 31, 64, 90, 144, 240]))
    n = int(rng.integers(1, 4))
    rgba = bool(rng.integers(0, 2))
    seqs = [synth.generate_sequence(w, h, 5, seed=int(rng.integers(1 << 30)), rgba=rgba,
                                    profile=str(rng.choice(["typical", "dense"])), raw_fraction=float(rng.choice([0.0, 0.3])))
            for _ in range(n)]
    ref = oracle.OracleStore(w, h, n)
    plain, staged = abi.VideoStore(hip_ctx, w, h, n), abi.VideoStore(hip_ctx, w, h, n)
    for step in range(5):
        pictures, merged = [], ([], [], [])
        mb0 = c0 = 0
        for sidx, seq in enumerate(seqs):
            s = seq[step]
            p, m = s.pics.copy(), s.mbs.copy()
            p["stream"] = sidx
            pictures.append((p[0], s.mbs, s.coefs))
            q = p.copy()
            q["mb_first"] = 0
            ref.submit(q, s.mbs, s.coefs)
            p["mb_first"] = mb0
            m["pic"] = sidx
            m["coef_off"] += c0
            merged[0].append(p)
            merged[1].append(m)
            merged[2].append(s.coefs)
            mb0 += len(m)
            c0 += len(s.coefs) // desc.COEF_UNIT
        plain.submit(np.concatenate(merged[0]), np.concatenate(merged[1]), np.concatenate(merged[2]))
        staged.submit_staged(pictures, threads=2)
        for sidx in range(n):
            for slot in range(3):
                want = ref.read_planes(sidx, slot)
                assert_planes_equal(want, plain.read_planes(sidx, slot), "plain: step %d stream %d slot %d" % (step, sidx, slot))
                assert_planes_equal(want, staged.read_planes(sidx, slot), "staged: step %d stream %d slot %d" % (step, sidx, slot))
            if rgba:
                cur = int(seqs[sidx][step].pics[0]["cur"])
                want = ref.read_rgba(sidx, cur).reshape(-1)
                assert np.array_equal(np.asarray(plain.read_rgba(sidx, cur)).reshape(-1), want)
                assert np.array_equal(np.asarray(staged.read_rgba(sidx, cur)).reshape(-1), want)
    plain.close()
    staged.close()


@pytest.mark.gpu
def test_stage_protocol_errors(hip_ctx):
    """One stage at a time; no plain submit while it is open; a commit with a picture missing launches nothing;
    every error leaves the store usable."""
    import ctypes as C
    w, h = 96, 80
    s = synth.generate_sequence(w, h, 2, seed=11)[0]
    dut = abi.VideoStore(hip_ctx, w, h, 2)
    L = dut.lib
    pics, mbs, coefs = dut._args(s.pics, s.mbs, s.coefs)
    n_mbs = np.array([len(mbs)] * 2, np.uint32)
    nbytes = np.array([coefs.nbytes] * 2, np.uint64)
    st = C.c_void_p()
    assert L.mpeghip_video_stage_begin(dut.h, 2, n_mbs.ctypes.data, nbytes.ctypes.data, C.byref(st)) == 0
    st2 = C.c_void_p()
    assert L.mpeghip_video_stage_begin(dut.h, 2, n_mbs.ctypes.data, nbytes.ctypes.data, C.byref(st2)) != 0
    assert b"still open" in L.mpeghip_last_error()
    with pytest.raises(abi.MpegHipError, match="stage is open"):
        dut.submit(s.pics, s.mbs, s.coefs)
    assert L.mpeghip_video_stage_put(st, 5, pics.ctypes.data, mbs.ctypes.data, coefs.ctypes.data) != 0   # no such picture
    st_err = L.mpeghip_video_stage_commit(st)
    assert st_err != 0 and b"picture 5 of 2" in L.mpeghip_last_error()
    assert L.mpeghip_video_stage_begin(dut.h, 2, n_mbs.ctypes.data, nbytes.ctypes.data, C.byref(st)) == 0
    assert L.mpeghip_video_stage_put(st, 0, pics.ctypes.data, mbs.ctypes.data, coefs.ctypes.data) == 0
    assert L.mpeghip_video_stage_commit(st) != 0 and b"never put" in L.mpeghip_last_error()
    cur = int(s.pics[0]["cur"])
    assert not dut.read_planes(0, cur)[0].any()          # nothing was launched
    dut.submit(s.pics, s.mbs, s.coefs)                    # and the store goes on working
    assert dut.read_planes(0, cur)[0].any()
    # an empty stage is legal
    assert L.mpeghip_video_stage_begin(dut.h, 0, None, None, C.byref(st)) == 0
    assert L.mpeghip_video_stage_commit(st) == 0
    dut.close()


def test_streams_are_independent(oracle, hip_ctx):
    """Different streams decode different pictures in ONE submit without interfering."""
    w, h = 96, 80
    seqs = [synth.generate_sequence(w, h, 4, seed=100 + i) for i in range(3)]
    ref, dut = oracle.OracleStore(w, h, 3), abi.VideoStore(hip_ctx, w, h, 3)
    for step in range(4):
        pics, mbs, coefs, mb0, c0 = [], [], [], 0, 0
        for sidx, seq in enumerate(seqs):
            s = seq[step]
            p, m = s.pics.copy(), s.mbs.copy()
            p["stream"], p["mb_first"] = sidx, mb0
            m["pic"] = sidx
            m["coef_off"] += c0
            pics.append(p)
            mbs.append(m)
            coefs.append(s.coefs)
            mb0 += len(m)
            c0 += len(s.coefs) // desc.COEF_UNIT
        pics, mbs, coefs = np.concatenate(pics), np.concatenate(mbs), np.concatenate(coefs)
        ref.submit(pics, mbs, coefs)
        dut.submit(pics, mbs, coefs)
        for sidx in range(3):
            for slot in range(3):
                assert_planes_equal(ref.read_planes(sidx, slot), dut.read_planes(sidx, slot), "stream %d slot %d" % (sidx, slot))
    dut.close()
