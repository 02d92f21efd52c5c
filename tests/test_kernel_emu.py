"""CPU-only check of the GPU kernels' lane logic: the lane functions the HIP
kernels are built from (mpeg_amd/csrc/*_lane.h), run by the test-only emulator,
must reproduce the oracle bit for bit.  (The GPU itself is checked by the
-m gpu tests through the C ABI; this catches addressing / lane-map errors here.)"""
import numpy as np
import pytest

from mpeg_amd import desc, synth
from parity import bits_equal, mirror_ring, run_and_compare


@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 10, "typical", 0.0, False),   # SIF, config 2
    (352, 240, 5, "typical", 0.15, False),   # with int32 snapshot blocks
    (352, 240, 4, "dense", 0.0, False),      # worst case: every block full, odd vectors
    (160, 120, 7, "typical", 0.05, True),    # fused RGBA
    (176, 144, 4, "typical", 0.0, True),     # QCIF: height not a multiple of 16 in RGBA
    (24, 40, 4, "typical", 0.0, True),       # tiny, width not a multiple of 16
    (50, 35, 3, "typical", 0.0, True),       # odd height: the last RGBA row has no partner
])
def test_video_lane_logic_matches_oracle(oracle, emu, w, h, n, profile, raw, rgba):
    seq = synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba)
    run_and_compare(oracle.OracleStore(w, h), emu.EmuStore(w, h), seq, check_rgba=rgba)


def test_video_custom_quant_matrices(oracle, emu):
    rng = np.random.default_rng(5)
    iq, nq = rng.integers(1, 256, 64), rng.integers(1, 256, 64)
    o, e = oracle.OracleStore(64, 48), emu.EmuStore(64, 48)
    o.set_quant(0, iq, nq)
    e.set_quant(0, iq, nq)
    run_and_compare(o, e, synth.generate_sequence(64, 48, 6, seed=11))


@pytest.mark.parametrize("w,h", [(100, 60), (37, 23), (16, 1), (3, 16)])  # odd sizes: partial quads, a last row without a partner
def test_standalone_rgba(oracle, emu, w, h):
    o, e = oracle.OracleStore(w, h), emu.EmuStore(w, h)
    rng = np.random.default_rng(3)
    g = desc.geometry(w, h)
    y, cb, cr = (rng.integers(0, 256, n, dtype=np.uint8) for n in (g["luma_bytes"], g["chroma_bytes"], g["chroma_bytes"]))
    o.write_planes(0, 1, y, cb, cr)
    e.write_planes(0, 1, y, cb, cr)
    e.rgba_convert(1)
    assert np.array_equal(o.read_rgba(0, 1), e.read_rgba(0, 1))


def test_rgba_arrangement_equals_the_reference_form_for_every_input(emu):
    """chroma terms + v_perm/v_sat_pk style clamp (video_lane.h rgba_pixel) == Go's DrawYCbCr arithmetic for all 2^24 inputs."""
    L = emu.lib()
    L.emu_rgba_forms_disagree.restype = __import__("ctypes").c_uint32
    assert L.emu_rgba_forms_disagree() == 0


def test_avg4_identity(emu):
    """(a+b+c+d+2)>>2 == ceil_avg(floor_avg(a,b), floor_avg(c,d)) + correction, for every pair of pair-sums."""
    L = emu.lib()
    for a in range(0, 256, 5):
        for b in (0, 1, 127, 128, 254, 255, a):
            for c in range(0, 256, 7):
                for d in (0, 1, 2, 129, 255, c):
                    want = (a + b + c + d + 2) >> 2
                    got = L.emu_avg4(a * 0x01010101, b * 0x01010101, c * 0x01010101, d * 0x01010101)
                    assert got == want * 0x01010101
    rng = np.random.default_rng(0)
    for _ in range(2000):
        a, b, c, d = (int(x) for x in rng.integers(0, 2**32, 4))
        got = L.emu_avg4(a, b, c, d)
        for k in range(4):
            s = sum((v >> (8 * k)) & 0xff for v in (a, b, c, d))
            assert (got >> (8 * k)) & 0xff == (s + 2) >> 2


def test_transposition_through_lds_in_halves(emu):
    """rc_tpose_store / rc_tpose_load (the instance for dense units): lane (g, j) enters with column j of block g and leaves with row
    j; half a wave at a time fits 1 152 bytes; the 32 lanes of each of the 8 store instructions fall on 32 different LDS banks
    (dword index mod 32 — and mod 64), and no two stores of a half share a dword."""
    import ctypes as C
    rng = np.random.default_rng(4)
    v = rng.integers(-2 ** 31, 2 ** 31, size=(64, 8), dtype=np.int64).astype(np.int32)
    want = v.reshape(8, 8, 8).transpose(0, 2, 1).reshape(64, 8).copy()   # [g][j][r] -> [g][r as lane][j as register]
    got = v.copy()
    touched = np.zeros(4 * 72, np.uint8)
    lib = emu.lib()
    lib.emu_tpose.restype = C.c_uint32
    assert lib.emu_tpose(got.ctypes.data_as(C.c_void_p), touched.ctypes.data_as(C.c_void_p)) == 1152
    assert np.array_equal(got, want)
    assert (touched != 0).sum() == 32 * 8                                # 256 values of a half, 256 distinct dwords
    for r in range(8):
        idx = np.nonzero(touched == 1 + r)[0]
        assert len(idx) == 32 and len(set(idx % 32)) == 32 and len(set(idx % 64)) == 32


def test_xcd_chunk_is_a_permutation(emu):
    L = emu.lib()
    for g8 in (1, 2, 7, 8, 9, 128, 513):     # the grid is 8 * g8 workgroups (launch_batch rounds it up)
        got = [L.emu_xcd_chunk(b, g8) for b in range(8 * g8)]
        assert sorted(got) == list(range(8 * g8))
        for x in range(8):                   # XCD x (blocks x, x + 8, ...) walks ONE contiguous range, in order
            assert got[x::8] == list(range(x * g8, (x + 1) * g8))


@pytest.mark.parametrize("fma", [0, 1])
@pytest.mark.parametrize("fmt", [desc.AUDIO_F32N, desc.AUDIO_F32NLR, desc.AUDIO_F32, desc.AUDIO_S16])
def test_audio_lane_logic_matches_oracle(oracle, emu, fma, fmt):
    s = synth.audio_frames(2, 5)
    o, e = oracle.OracleSynth(2, fma), emu.EmuSynth(2, fma)
    for _ in range(3):  # state (V ring, vPos) carried across calls
        assert bits_equal(o.synth(s, fmt), e.synth(s, fmt))
        for st in range(2):
            (va, pa), (vb, pb) = o.get_state(st), e.get_state(st)
            assert pa == pb and bits_equal(va, vb)


@pytest.mark.parametrize("chunks", [1, 2, 3, 7])
def test_audio_time_slices_are_bit_identical(oracle, emu, chunks):
    """Splitting a launch along time (history rebuilt from the samples) must not change a single bit."""
    s = synth.audio_frames(2, 9)
    o, e = oracle.OracleSynth(2, 0), emu.EmuSynth(2, 0, chunks=chunks)
    for _ in range(2):
        assert bits_equal(o.synth(s, desc.AUDIO_F32N), e.synth(s, desc.AUDIO_F32N))
        for st in range(2):
            (va, pa), (vb, pb) = o.get_state(st), e.get_state(st)
            assert pa == pb and bits_equal(va, vb)


@pytest.mark.parametrize("scale", [2.0 ** -125, 2.0 ** -140, 1.0])
def test_audio_scaling_of_tiny_sums_takes_the_long_division(oracle, emu, scale):
    """Sums inside 2^-119..2^-95 are where the short division is not proven (tests/proofs); a
    history of tiny values and silent samples puts every output there."""
    rng = np.random.default_rng(5)
    v = mirror_ring(rng.integers(-999, 1000, (2, 16, 32)).astype(np.float32) * np.float32(scale))
    o, e = oracle.OracleSynth(1, 0), emu.EmuSynth(1, 0)
    o.set_state(0, v, 320)
    e.set_state(0, v, 320)
    s = np.zeros((1, 1, 2, 36, 32), np.int32)
    a, b = o.synth(s, desc.AUDIO_F32N), e.synth(s, desc.AUDIO_F32N)
    assert bits_equal(a, b) and np.count_nonzero(a) > 500
    if scale == 2.0 ** -125:
        mag = np.abs(a[a != 0]) * 1090519040.0
        assert ((mag > 2.0 ** -119) & (mag < 2.0 ** -95)).mean() > 0.5


@pytest.mark.parametrize("fma,want", [(0, 0xf1b76cdf8e6cdea5), (1, 0x50f3ab75f5fb0fb5)])
def test_audio_lane_logic_reproduces_golden_hash(oracle, emu, golden_dir, fma, want):
    """Real sub-band samples of test.mp2 (parsed by the oracle) through the kernel's lane functions."""
    dec = oracle.AudioDecoder((golden_dir / "test.mp2").read_bytes(), fma)
    frames = []
    while True:
        r = dec.decode(True)
        if r is None:
            break
        frames.append(r[1])
    S = np.stack(frames)[None]
    e, h, i = emu.EmuSynth(1, fma), oracle.FNV_OFFSET, 0
    for chunk in (1, 2, 5, 17, 100, 1000):
        part = S[:, i:i + chunk]
        if part.shape[1]:
            h = oracle.fnv1a64(e.synth(part, 0), h)
        i += chunk
    assert h == want


@pytest.mark.parametrize("profile,raw", [("typical", 0.0), ("typical", 0.2), ("dense", 0.0)])
def test_packer_512_bit_forms_write_the_same_words(emu, profile, raw):
    """rc_pack_picture with and without its AVX-512 forms (entries by widen + compress instead of a bit-scan loop):
    identical chunks and words, so a host without AVX-512 feeds the kernel the same bytes."""
    import ctypes as C
    L = emu.lib()
    if not L.emu_host_has_avx512():
        pytest.skip("this CPU has no AVX-512: only the narrow form exists here")
    g = desc.geometry(176, 144)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    for s in synth.generate_sequence(176, 144, 4, profile=profile, raw_fraction=raw, seed=77):
        pics = np.ascontiguousarray(s.pics, desc.PIC_DTYPE)
        mbs = np.ascontiguousarray(s.mbs, desc.MB_DTYPE)
        coefs = np.ascontiguousarray(s.coefs).view(np.uint8).reshape(-1)
        got = []
        for f in (L.emu_pack, L.emu_pack_narrow):
            chunks = np.zeros((len(mbs) // 4 + 4) * emu.CHUNK_DWORDS, np.uint32)
            words = np.zeros(len(coefs) // 2 + len(mbs) * 6 + 1024, np.uint32)
            nw = C.c_uint32(0)
            nc = f(g["luma_w"], g["luma_h"], 1 << 20, 1 << 20, P(pics), P(mbs), P(coefs), P(chunks), P(words), C.byref(nw))
            got.append((nc, nw.value, chunks[:nc * emu.CHUNK_DWORDS].copy(), words[:nw.value].copy()))
        assert got[0][0] == got[1][0] and got[0][1] == got[1][1]
        assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3])


@pytest.mark.parametrize("chunks", [1, 3])
def test_audio_step_grid_at_every_ring_position(oracle, emu, chunks):
    """The kernel's steps start where the window's ring position is 15, i.e. up to 15 sub-blocks BEFORE a slice
    (audio_step_base0): every one of the 16 alignments, with the state's own history in front of the first slice and a
    rebuilt one in front of the others."""
    rng = np.random.default_rng(11)
    s = synth.audio_frames(1, 4, seed=3)
    for vpos in range(0, 1024, 64):
        v = mirror_ring(rng.integers(-30000, 30000, (2, 16, 32)).astype(np.float32))
        o, e = oracle.OracleSynth(1, 0), emu.EmuSynth(1, 0, chunks=chunks)
        o.set_state(0, v, vpos)
        e.set_state(0, v, vpos)
        assert bits_equal(o.synth(s, desc.AUDIO_F32N), e.synth(s, desc.AUDIO_F32N)), vpos
        (va, pa), (vb, pb) = o.get_state(0), e.get_state(0)
        assert pa == pb and bits_equal(va, vb), vpos


def test_audio_slices_tile_the_launch_on_the_step_grid(emu):
    """audio_slice_range: the non-empty slices of a stream are disjoint, in order and cover [0, n_frames * 36); every
    cut but 0 and the end is a whole number of 32-sub-block steps away from the stream's first grid point
    (vpos0 / 64 mod 16), and lies at least 15 sub-blocks in (the history rebuilt in front of a slice)."""
    import ctypes as C
    L = emu.lib()
    for n_frames in (1, 4, 9, 20, 100):
        for n_chunks in (1, 2, 3, 5, 7, 25):
            if n_chunks > n_frames:
                continue
            for vpos0 in range(0, 1024, 64):
                g, n, at = (vpos0 >> 6) & 15, n_frames * 36, 0
                for chunk in range(n_chunks):
                    t0, t1 = C.c_uint32(0), C.c_uint32(0)
                    L.emu_audio_slice_range(n_frames, n_chunks, vpos0, chunk, C.byref(t0), C.byref(t1))
                    if t0.value >= t1.value:
                        continue
                    assert t0.value == at, (n_frames, n_chunks, vpos0, chunk)
                    if t0.value:
                        assert t0.value >= 15 and (t0.value - g) % 32 == 0
                    at = t1.value
                assert at == n, (n_frames, n_chunks, vpos0)


def test_a_window_that_ends_with_the_slot_reads_nothing_behind_the_allocation(oracle, emu):
    """Round 2's advisor finding: a kRSlow gather always fetches 17 luma rows x 32 bytes / 9 chroma rows x 16 bytes from the
    dword below the window origin — a row and a piece more than the half-pel mode (and the validation) needs.  A vector the
    reference accepts — bottom-row macroblock, mv_y = +128: its Cr window ends with the slot's pad — makes that extra row run
    up to chroma_w + 8 bytes past frame_bytes; for slot 2 of the last stream that is past the frame store unless the store
    has its tail pad (mpeghip_video_open).  Here the store ends exactly at an inaccessible page, sized like the product's."""
    w, h, n = 640, 208, 2          # chroma_w + 8 = 328 bytes of overrun > the stride's own slack (at most 319)
    g = desc.geometry(w, h)
    assert g["luma_h"] // 4 + 16 > 64          # the chroma planes' reach into the pad is what bounds the vector, not luma's
    ref, dut = oracle.OracleStore(w, h, n), emu.EmuStore(w, h, n, guard=True)
    rng = np.random.default_rng(4)
    for st in range(n):
        for slot in range(3):
            y, cb, cr = (rng.integers(0, 256, k, dtype=np.uint8) for k in (g["luma_bytes"], g["chroma_bytes"], g["chroma_bytes"]))
            ref.write_planes(st, slot, y, cb, cr)
            dut.write_planes(st, slot, y, cb, cr)
    mbs = np.zeros(g["mb_w"], desc.MB_DTYPE)
    mbs["mb_x"], mbs["mb_y"] = np.arange(g["mb_w"]), g["mb_h"] - 1
    mbs["mv_x"], mbs["mv_y"] = 0, 128
    mbs["flags"] = desc.MB_REF_FWD
    mbs["qscale"] = 1
    pics = np.zeros(1, desc.PIC_DTYPE)
    pics["stream"], pics["cur"], pics["fwd"], pics["bwd"] = n - 1, 0, 2, 1     # predicts from the LAST slot of the LAST stream
    pics["mb_count"] = len(mbs)
    coefs = np.zeros(0, np.uint8)
    ref.submit(pics, mbs, coefs)
    dut.submit(pics, mbs, coefs)          # (a read past the store would fault here)
    for a, b in zip(ref.read_planes(n - 1, 0), dut.read_planes(n - 1, 0)):
        assert np.array_equal(a, b)
