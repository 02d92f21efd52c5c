"""The two instances of the reconstruction kernel (video_recon_lane.h: int16 coefficient tile in both; transposition across lanes,
or through LDS in two halves + the short dequantisation of dense units; the library picks one per batch,
mpeghip_video_set_tile_policy pins one) keep the
reference's results bit for bit on the same cases: prediction windows, windows that leave their plane (the linear reads),
stores of runs and of single macroblocks, the fused and the whole-frame RGBA, snapshot blocks, dense units."""
import numpy as np
import pytest

from mpeg_amd import desc, synth
from parity import run_and_compare

TILES = [("int16", 1), ("int32", 2)]


@pytest.fixture(params=TILES, ids=[t for t, _ in TILES])
def emu_layout(request, emu):
    emu.set_tile_policy(request.param[1])
    yield emu
    emu.set_tile_policy(0)


@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 6, "typical", 0.0, False),
    (352, 240, 4, "typical", 0.15, False),
    (352, 240, 3, "dense", 0.0, False),
    (160, 120, 5, "typical", 0.05, True),
    (176, 144, 4, "typical", 0.0, True),
    (24, 40, 4, "typical", 0.0, True),
    (50, 35, 3, "typical", 0.0, True),
])
def test_layout_option_matches_oracle(oracle, emu_layout, w, h, n, profile, raw, rgba):
    seq = synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba)
    run_and_compare(oracle.OracleStore(w, h), emu_layout.EmuStore(w, h), seq, check_rgba=rgba)


@pytest.mark.parametrize("w,h", [(100, 60), (37, 23), (16, 1), (3, 16)])
def test_layout_option_standalone_rgba_and_plane_round_trip(oracle, emu_layout, w, h):
    o, e = oracle.OracleStore(w, h), emu_layout.EmuStore(w, h)
    rng = np.random.default_rng(3)
    g = desc.geometry(w, h)
    y, cb, cr = (rng.integers(0, 256, n, dtype=np.uint8) for n in (g["luma_bytes"], g["chroma_bytes"], g["chroma_bytes"]))
    o.write_planes(0, 1, y, cb, cr)
    e.write_planes(0, 1, y, cb, cr)
    for a, b in zip((y, cb, cr), e.read_planes(0, 1)):
        assert np.array_equal(a, b)
    e.rgba_convert(1)
    assert np.array_equal(o.read_rgba(0, 1), e.read_rgba(0, 1))


def test_the_frame_store_keeps_cb_and_cr_of_a_macroblock_side_by_side(emu):
    """video_lane.h: Cb and Cr of macroblock i lie side by side behind the luma plane (128 bytes per macroblock = one cache
    line for both planes of a prediction window's macroblock)."""
    w, h = 64, 32
    g = desc.geometry(w, h)
    L, C = g["luma_bytes"], g["chroma_bytes"]
    y = np.zeros(L, np.uint8)
    cb, cr = np.full(C, 0xB0, np.uint8), np.full(C, 0xC0, np.uint8)
    cb[:8] = np.arange(8)                                        # row 0 of the first block of Cb
    e = emu.EmuStore(w, h)
    e.write_planes(0, 0, y, cb, cr)
    pairs = e.frames[L:L + 2 * C]
    assert list(pairs[:8]) == list(range(8))
    blocks = pairs.reshape(-1, 2, 64)
    assert (blocks[:, 1] == 0xC0).all() and (blocks[1:, 0] == 0xB0).all()      # Cb | Cr per macroblock


def test_any_intra_dc(oracle, emu_layout):
    """The int16 tile holds dequantised AC levels only (|.| <= 2048); an intra block's DC — any int16 through the ABI,
    `<< 8` in the reference (video.go:672) — rides in the block's word of the device format (sparse blocks) or is read
    from the unit (dense blocks)."""
    w, h = 64, 48
    seq = synth.generate_sequence(w, h, 4, seed=21)
    hit = 0
    for sub in seq:
        units = sub.coefs.view(np.int16).reshape(-1, 64)
        for i, mb in enumerate(sub.mbs):
            if (mb["flags"] & desc.MB_INTRA) and not (mb["flags"] & desc.MB_COEF_RAW) and mb["cbp"]:
                units[mb["coef_off"], 0] = (4096, -4096, 4095, -4095, 32767, -32768)[hit % 6]
                hit += 1
    assert hit >= 6
    run_and_compare(oracle.OracleStore(w, h), emu_layout.EmuStore(w, h), seq)


def test_levels_over_the_whole_int16_range(oracle, emu_layout):
    """The ABI takes any int16 level, any quantiser_scale 1..31 and any matrix bytes — far beyond what MPEG-1 codes
    (+-255): every third non-zero level of every block replaced by extremes, both kernels' arithmetic against the oracle."""
    rng = np.random.default_rng(5)
    for trial in range(4):
        w, h = 96, 64
        seq = synth.generate_sequence(w, h, 3, seed=100 + trial, profile="typical" if trial % 2 else "dense")
        for sub in seq:
            units = sub.coefs.view(np.int16).reshape(-1, 64)
            for u in range(len(units)):
                nz = np.nonzero(units[u])[0]
                if len(nz):
                    pick = rng.choice(nz, size=max(1, len(nz) // 3), replace=False)
                    units[u, pick] = rng.choice([32767, -32768, 2047, -2048, 256, -256, 1, -1, 12345, -23456], size=len(pick)).astype(np.int16)
            sub.mbs["qscale"] = rng.choice([1, 31, 17], size=len(sub.mbs))
        o, e = oracle.OracleStore(w, h), emu_layout.EmuStore(w, h)
        if trial >= 2:
            iq, nq = rng.integers(1, 256, 64), rng.integers(1, 256, 64)
            o.set_quant(0, iq, nq)
            e.set_quant(0, iq, nq)
        run_and_compare(o, e, seq)


def test_dense_units_with_zeros_large_levels_and_extreme_matrices(oracle, emu_layout):
    """The dense path works on packed 16-bit halves (2 * quantiser_scale * level + quantiser_scale must fit: the packer keeps
    levels beyond +-528 out of it), takes a zero level through the whole chain as 0, and multiplies by any matrix byte, 0 and 1
    included."""
    rng = np.random.default_rng(9)
    for trial in range(4):
        w, h = 96, 64
        seq = synth.generate_sequence(w, h, 3, seed=300 + trial, profile="dense")
        n_dense = 0
        for sub in seq:
            units = sub.coefs.view(np.int16).reshape(-1, 64)
            for u in range(len(units)):
                pick = rng.choice(64, size=24, replace=False)
                beyond = [529, -529, 16384, -16384] if u % 5 == 0 else []   # (one unit in five: beyond the dense path -> entries)
                units[u, pick] = rng.choice([528, -528, 527, -500, 255, -255, 1, -1, 0, 0, 0] + beyond,
                                            size=len(pick)).astype(np.int16)
                n_dense += int(np.count_nonzero(units[u]) > 32 and np.abs(units[u, 1:].astype(np.int32)).max() <= 528)
            sub.mbs["qscale"] = rng.choice([1, 2, 31, 17], size=len(sub.mbs))
        assert n_dense > 100
        o, e = oracle.OracleStore(w, h), emu_layout.EmuStore(w, h)
        if trial >= 1:
            iq = rng.choice([0, 1, 2, 16, 255], size=64) if trial == 1 else rng.integers(0, 256, 64)
            nq = rng.choice([0, 1, 3, 16, 255], size=64) if trial == 1 else rng.integers(0, 256, 64)
            o.set_quant(0, iq, nq)
            e.set_quant(0, iq, nq)
        run_and_compare(o, e, seq)


@pytest.mark.parametrize("case", ["default", "custom_intra_default_non_intra", "first_entry_off", "last_entry_off", "all_17"])
def test_dense_units_take_the_short_dequantisation_only_under_the_default_non_intra_matrix(oracle, emu_layout, case):
    """rc_dense_cols<true>: with the non-intra matrix 16 everywhere (video.go:1066-1075) a pass whose dense units are all
    non-intra dequantises as u * quantiser_scale.  One entry off, anywhere, and the general path must run; intra units
    (the I picture) always take it."""
    rng = np.random.default_rng(21)
    w, h = 96, 64
    seq = synth.generate_sequence(w, h, 4, seed=77, profile="dense")
    for sub in seq:
        sub.mbs["qscale"] = rng.integers(1, 32, size=len(sub.mbs))
    o, e = oracle.OracleStore(w, h), emu_layout.EmuStore(w, h)
    iq, nq = rng.integers(1, 256, 64), np.full(64, 16)
    if case == "first_entry_off":
        nq[0] = 17
    elif case == "last_entry_off":
        nq[63] = 15
    elif case == "all_17":
        nq[:] = 17
    if case != "default":
        o.set_quant(0, iq, nq)
        e.set_quant(0, iq, nq)
    run_and_compare(o, e, seq)


def test_the_emulator_picks_the_instance_by_the_library_s_rule():
    """launch_batch (mpeghip.hip) and the emulator's copy of its rule name the same share of dense blocks (the emulator's choice
    decides which lane functions the CPU suite runs under the automatic policy)."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    hip = (root / "mpeg_amd" / "csrc" / "mpeghip.hip").read_text()
    emu_src = (root / "tests" / "kernel_emu" / "emu.cpp").read_text()
    num, den = map(int, re.search(r"kDenseShareNum = (\d+), kDenseShareDen = (\d+);", hip).groups())
    assert "b->dense_blocks * kDenseShareDen <= b->coded_blocks * kDenseShareNum" in hip
    m = re.search(r"bool t16 = dense \* (\d+) <= coded( \* (\d+))?;", emu_src)
    assert m and int(m.group(1)) == den and int(m.group(3) or 1) == num


@pytest.fixture
def emu_wide(emu):
    emu.set_wide(1)
    yield emu
    emu.set_wide(0)


@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 6, "typical", 0.0, False),     # SIF: chunks that wrap a row end are runs (luma by wave 0, chroma by wave 1)
    (352, 240, 4, "typical", 0.15, True),     # snapshot blocks; fused RGBA: four image rows per wave / macroblock w of a wrapped run
    (352, 240, 3, "dense", 0.0, False),       # dense units through rc_dense_cols<false>, three passes = three waves with a tile each
    (160, 120, 5, "typical", 0.05, True),
    (176, 144, 4, "typical", 0.0, True),
    (24, 40, 4, "typical", 0.0, True),        # chunks with dead records: waves beyond the live macroblocks store nothing
    (50, 35, 3, "typical", 0.0, True),
])
def test_the_wide_kernels_four_wave_orchestration_matches_the_oracle(oracle, emu_wide, w, h, n, profile, raw, rgba):
    """recon_wide_kernel (what every launch of at most two 1080p pictures runs) on the CPU: its per-wave table and window loads,
    pass w on tile w, the two barriers' hand-over of the output bytes and the split stores, in the kernel's order over the same
    chunk format and lane functions (tests/kernel_emu/emu.cpp: emu_wide_chunk) — until round 6 only the -m gpu tests ran this
    kernel's orchestration (round-5 advisor)."""
    seq = synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba)
    run_and_compare(oracle.OracleStore(w, h), emu_wide.EmuStore(w, h), seq, check_rgba=rgba)


def test_the_wide_orchestration_equals_the_one_wave_form_on_the_golden_streams_pictures(oracle, emu, golden_dir):
    """... and through the host parser on the damaged golden stream (windows that leave their plane, invalid intra blocks, re-submits):
    the reference's hash with every chunk run the four-wave way."""
    import hostlib
    E = hostlib.host_emu()     # (the parser's test backend carries its own copy of the lane emulator: its switch, not emu's)
    E.emu_set_wide(1)
    try:
        dec = hostlib.HostVideo((golden_dir / "test.mpeg1video").read_bytes(), emu_flavour=0)
        h, n = oracle.FNV_OFFSET, 0
        while True:
            f = dec.decode()
            if f is None:
                break
            for p in hostlib.frame_planes(f):
                h = oracle.fnv1a64(p, h)
            n += 1
        dec.close()
        import ctypes as C
        E.emu_wide_chunks_run.restype = C.c_uint64
        assert E.emu_wide_chunks_run() >= 5000               # every chunk of the 261 pictures (20 each, but for the damage) went the four-wave way
    finally:
        E.emu_set_wide(0)
    assert (h, n) == (0xea6d7fcb1340ba3f, 260)
