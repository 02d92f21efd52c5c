"""The DEVICE-side packer (mpeg_amd/csrc/video_pack_lane.h: what pack_kernel runs in front of recon_kernel for a
device-packed stage, include/mpeghip.h: mpeghip_video_stage_begin_device) on the CPU, lane by lane through the lane emulator:
device-packed = host-packed (chunk for chunk, word for word) = oracle (frames), and every malformed picture the host packer
refuses is reported by the device packer too.  The GPU twin is tests/test_gpu_device_pack.py."""
import numpy as np
import pytest

from mpeg_amd import desc, synth
from parity import assert_planes_equal

NO_ERROR = 0xFFFFFFFFFFFFFFFF
REASONS = {1: "position", 2: "refs", 3: "cbp", 4: "qscale", 5: "same slot", 6: "range", 7: "twice", 8: "sparse", 9: "order", 10: "depends"}


def _geom(emu, w, h):
    st = emu.EmuStore(w, h)
    return st.g, st.stride, st.rgba_stride


def _compare_packed(host, dev_chunks, dev_words, word_first=0):
    """Host-packed and device-packed forms of one picture: the chunks equal but for where their words begin, and each chunk's
    words (block words, entries, snapshot / dense data) equal word for word."""
    hc, hw = host
    assert hc.shape == dev_chunks.shape
    for c in range(len(hc)):
        a, b = hc[c].copy(), dev_chunks[c].copy()
        n_slots = int(a[5] & 0x1f)
        # h2 h3: the byte offset (64 bits) of the chunk's first block word
        a3, b3 = (int(a[2]) | int(a[3]) << 32) // 4, (int(b[2]) | int(b[3]) << 32) // 4
        a[2] = a[3] = b[2] = b[3] = 0
        assert (a == b).all(), "chunk %d: %s vs %s" % (c, a, b)
        # the chunk's extent: block words + entries + the data behind them (where the last snapshot / dense block ends)
        ne = sum((int(a[4]) >> (10 * p)) & 0x3ff for p in range(3))
        extent = n_slots + ne
        for s in range(n_slots):
            bw = int(hw[a3 + s])
            if bw & (1 << 10):      # snapshot
                extent = max(extent, n_slots + ((bw >> 12) & 0x3fff) + 64)
            elif bw & (1 << 11):    # dense unit
                extent = max(extent, n_slots + ((bw >> 12) & 0x3fff) + 32)
        assert b3 >= word_first
        assert np.array_equal(hw[a3:a3 + extent], dev_words[b3:b3 + extent]), "chunk %d: words differ" % c


@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 5, "typical", 0.0, False),
    (352, 240, 4, "typical", 0.2, True),    # snapshot blocks
    (352, 240, 3, "dense", 0.0, False),     # every block beyond 32 levels: units built from the pairs on the device
    (160, 120, 5, "typical", 0.05, True),
    (50, 35, 3, "typical", 0.1, False),     # a last chunk with dead records, a last wave with idle lanes
    (1920, 1080, 2, "typical", 0.01, False),
])
def test_device_packed_equals_host_packed(emu, w, h, n, profile, raw, rgba):
    g, stride, rgba_stride = _geom(emu, w, h)
    for i, s in enumerate(synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba, seed=0xD0 + n)):
        mbs, words = desc.to_sparse(s.mbs, s.coefs)
        pic = s.pics[0].copy()
        pic["stream"] = 0
        host = emu.pack_sparse_host(g, stride, rgba_stride, pic, mbs, words)
        assert host is not None
        for word_first, chunk_first in ((0, 0), (48, 7)):
            err, dc, dw, use = emu.pack_sparse_device(g, stride, rgba_stride, pic, mbs, words, word_first, chunk_first)
            assert err == NO_ERROR, "picture %d: %s at macroblock %d" % (i, REASONS.get(err & 0xff), err >> 8)
            _compare_packed(host, dc, dw, word_first)
            assert (dw[:word_first] == 0xDEADBEEF).all()
            want_use = (1 if (s.mbs["flags"] & desc.MB_REF_FWD).any() else 0) | (2 if (s.mbs["flags"] & desc.MB_REF_BWD).any() else 0)
            assert use == want_use


@pytest.mark.parametrize("window", [0, 40, 333, 4096])
def test_the_lds_window_changes_nothing(emu, window):
    """pack_kernel reads a wave's words through an LDS window of 4 096 dwords and from memory beyond it: with windows that end
    inside a macroblock's data, inside a block, or are not there at all, the packed form is the same."""
    w, h = 176, 144
    g, stride, rgba_stride = _geom(emu, w, h)
    emu.set_pack_window(window)
    try:
        for profile, raw in (("typical", 0.1), ("dense", 0.0)):
            for s in synth.generate_sequence(w, h, 3, profile=profile, raw_fraction=raw, seed=0x31):
                mbs, words = desc.to_sparse(s.mbs, s.coefs)
                host = emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, words)
                err, dc, dw, _ = emu.pack_sparse_device(g, stride, rgba_stride, s.pics[0], mbs, words, 16, 2)
                assert err == NO_ERROR
                _compare_packed(host, dc, dw, 16)
    finally:
        emu.set_pack_window(4096)


def test_device_packed_pictures_reconstruct_like_the_oracle(oracle, emu):
    """The whole path on the CPU: device packer's lanes -> reconstruction kernel's lanes, against the oracle; both kernel instances,
    a stream other than 0, coded zero levels and sparse gaps between macroblocks' data."""
    w, h = 176, 144
    seq = synth.generate_sequence(w, h, 6, profile="typical", raw_fraction=0.1, rgba=True, seed=77)
    emu.set_device_pack(1)
    try:
        for tile in (1, 2):
            emu.set_tile_policy(tile)
            ref, dut = oracle.OracleStore(w, h), emu.EmuStore(w, h)
            for s in seq:
                ref.submit(s.pics, s.mbs, s.coefs)
                mbs, words = desc.to_sparse(s.mbs, s.coefs)
                # gaps: every macroblock's data moved back by its index (offsets stay in order, words in between are unused)
                gap = np.zeros(len(words) + len(mbs) * 2, np.uint32) + np.uint32(0x7fffffff)
                m2 = mbs.copy()
                for k in range(len(mbs)):
                    a = int(mbs[k]["coef_off"])
                    b = int(mbs[k + 1]["coef_off"]) if k + 1 < len(mbs) else len(words)
                    gap[a + 2 * k:b + 2 * k] = words[a:b]
                    m2[k]["coef_off"] = a + 2 * k
                assert dut.submit_sparse(s.pics[0], m2, gap) == 0
                for slot in range(3):
                    assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(0, slot), "device-packed vs oracle, slot %d" % slot)
                assert np.array_equal(ref.read_rgba(0, s.cur), dut.read_rgba(0, s.cur))
    finally:
        emu.set_tile_policy(0)
        emu.set_device_pack(0)


def _first_mb(mbs, cond):
    return next(i for i, m in enumerate(mbs) if cond(m))


def _damage_cases():
    def count(m, w, k):
        w[int(m[k]["coef_off"])] = 65

    def stray(m, w, k):
        w[int(m[k]["coef_off"]) + 1] |= 0x0100

    def stray_low(m, w, k):
        w[int(m[k]["coef_off"]) + 2] |= 0x2

    def short(m, w, k):
        del w[int(m[k]["coef_off"]) + 3:]

    def overlap(m, w, k):     # two macroblocks naming the same words (the advisor's round-3 finding: a heap overrun on the host)
        m[k + 1]["coef_off"] = m[k]["coef_off"]

    def all_zero_offsets(m, w, k):
        m["coef_off"] = 0

    def beyond(m, w, k):
        m[k]["coef_off"] = len(w) + 1

    def position(m, w, k):
        m[k]["mb_x"] = 200

    def twice(m, w, k):
        m[k]["mb_x"], m[k]["mb_y"] = m[k - 1]["mb_x"], m[k - 1]["mb_y"]

    def cbp(m, w, k):
        m[k]["cbp"] = 0x40

    def qscale(m, w, k):
        m[k]["qscale"] = 0

    def refs(m, w, k):
        m[k]["flags"] |= desc.MB_REF_FWD | desc.MB_REF_BWD

    def vector(m, w, k):
        m[k]["mv_y"] = -2000

    return [("count", count, 8, True), ("stray", stray, 8, True), ("stray_low", stray_low, 8, True), ("short", short, 8, True),
            ("overlap", overlap, 9, True), ("all_zero_offsets", all_zero_offsets, 9, True), ("beyond", beyond, 8, True),
            ("position", position, 1, False), ("twice", twice, 7, False), ("cbp", cbp, 3, False), ("qscale", qscale, 4, False),
            ("refs", refs, 2, False), ("vector", vector, 6, False)]


@pytest.mark.parametrize("name,damage,reason,host_too", _damage_cases(), ids=[c[0] for c in _damage_cases()])
def test_malformed_pictures_are_reported(emu, name, damage, reason, host_too):
    """Every kind of malformed input: the device packer reports it (and the reason the host's validation would give), writes
    nothing outside its picture, and the host packer — where the check is the packer's, not validate_mb's — refuses it too."""
    w, h = 96, 64
    g, stride, rgba_stride = _geom(emu, w, h)
    seq = synth.generate_sequence(w, h, 2, seed=8, profile="dense")
    s = seq[1]     # a P picture: predicted macroblocks with coded blocks
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    words = [int(x) for x in words]
    k = _first_mb(mbs[1:-1], lambda m: m["cbp"] and not (m["flags"] & (desc.MB_COEF_RAW | desc.MB_INTRA))) + 1
    damage(mbs, words, k)
    words = np.array(words, np.uint32)
    err, dc, dw, _ = emu.pack_sparse_device(g, stride, rgba_stride, s.pics[0], mbs, words, 32, 3)
    assert err != NO_ERROR and (err & 0xff) == reason, (name, REASONS.get(err & 0xff), err >> 8)
    if name not in ("all_zero_offsets", "twice"):
        assert (err >> 8) in (k, k + 1), (name, err >> 8, k)
    if host_too:
        assert emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, words) is None, name


def test_an_intra_block_needs_its_dc_first_on_the_device_too(emu):
    w, h = 48, 32
    g, stride, rgba_stride = _geom(emu, w, h)
    s = synth.generate_sequence(w, h, 1, seed=8)[0]
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    k = _first_mb(mbs, lambda m: (m["flags"] & desc.MB_INTRA) and not (m["flags"] & desc.MB_COEF_RAW) and m["cbp"])
    words = words.copy()
    words[int(mbs[k]["coef_off"]) + 1] |= 5 << 2
    err, _, _, _ = emu.pack_sparse_device(g, stride, rgba_stride, s.pics[0], mbs, words)
    assert (err & 0xff) == 8 and (err >> 8) == k
    assert emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, words) is None


def test_a_snapshot_block_carries_its_count_word(emu):
    """ABI version 2: every block of the sparse form begins with a count word — 64 for a snapshot block."""
    w, h = 64, 48
    g, stride, rgba_stride = _geom(emu, w, h)
    s = synth.generate_sequence(w, h, 1, seed=3, raw_fraction=0.5)[0]
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    k = _first_mb(mbs, lambda m: (m["flags"] & desc.MB_COEF_RAW) and m["cbp"])
    at = int(mbs[k]["coef_off"])
    assert words[at] == 64
    bad = words.copy()
    bad[at] = 63
    assert emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, bad) is None
    err, _, _, _ = emu.pack_sparse_device(g, stride, rgba_stride, s.pics[0], mbs, bad)
    assert (err & 0xff) == 8 and (err >> 8) == k


def test_the_host_packer_checks_its_room(emu):
    """Pictures of one submit may name the same words; the packer is told how much room is left and refuses a picture that would
    not fit instead of writing past the buffer."""
    w, h = 96, 64
    g, stride, rgba_stride = _geom(emu, w, h)
    s = synth.generate_sequence(w, h, 1, seed=8, profile="dense")[0]
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    assert emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, words) is not None
    assert emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, words, out_room=len(words) // 2) is None


def test_mutated_pictures_host_and_device_agree(emu):
    """Random damage to descriptors and words: the two packers accept and refuse the same pictures (where the host's refusal is
    the packer's own — descriptor fields are validate_mb's business on the host), and accepted pictures pack identically."""
    import random
    w, h = 64, 48
    g, stride, rgba_stride = _geom(emu, w, h)
    s = synth.generate_sequence(w, h, 2, seed=17, profile="typical", raw_fraction=0.1)[1]
    mbs0, words0 = desc.to_sparse(s.mbs, s.coefs)
    agreed_bad = agreed_ok = 0
    for seed in range(200):
        rng = random.Random(seed)
        mbs, words = mbs0.copy(), words0.copy()
        for _ in range(rng.randrange(1, 4)):
            if rng.random() < 0.7:
                words[rng.randrange(len(words))] ^= np.uint32(1 << rng.randrange(32))
            else:
                mbs[rng.randrange(len(mbs))]["coef_off"] ^= np.uint32(1 << rng.randrange(12))
        host = emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, words)
        err, dc, dw, _ = emu.pack_sparse_device(g, stride, rgba_stride, s.pics[0], mbs, words, 16, 1)
        assert (host is None) == (err != NO_ERROR), (seed, REASONS.get(err & 0xff), err >> 8)
        if host is None:
            agreed_bad += 1
        else:
            _compare_packed(host, dc, dw, 16)
            agreed_ok += 1
    assert agreed_bad >= 30 and agreed_ok >= 30, (agreed_bad, agreed_ok)


K_CRUN = 1 << 30   # video_recon_lane.h: kCRun (header h4)


def _chunk_positions(chunks):
    """-> per chunk the (mb_x, mb_y) of its live records (record dword 0: mb_x << 16 | mb_y << 24; kRDead = 2)"""
    out = []
    for c in chunks:
        recs = [int(c[8 + 6 * m]) for m in range(4)]
        out.append([((r >> 16) & 0xff, r >> 24) for r in recs if not (r & 2)])
    return out


@pytest.mark.parametrize("w,h", [(352, 240), (160, 120), (176, 144)])   # mb_w = 22, 10, 11: not multiples of the chunk's 4
@pytest.mark.parametrize("rgba", [False, True])
def test_a_chunk_that_wraps_a_row_end_is_a_run(oracle, emu, w, h, rgba):
    """Tiles lie in raster order in the frame store, so 4 macroblocks consecutive in raster order are 4 consecutive tiles also when
    the chunk starts at the end of one macroblock row and ends in the next: both packers flag it kCRun (rc_run_follows — until
    round 6 only chunks inside one row were runs; at SIF one chunk in 5.5 took the per-block stores), the planes leave as 1 024 +
    512 contiguous bytes, and the fused colour conversion of such a chunk goes macroblock by macroblock (rc_run_in_one_row)."""
    g, stride, rgba_stride = _geom(emu, w, h)
    mb_w = g["luma_w"] // 16
    seq = synth.generate_sequence(w, h, 4, profile="typical", rgba=rgba, seed=0x77)
    wrapped_runs = 0
    for s in seq:
        mbs, words = desc.to_sparse(s.mbs, s.coefs)
        host = emu.pack_sparse_host(g, stride, rgba_stride, s.pics[0], mbs, words)
        err, dc, dw, _ = emu.pack_sparse_device(g, stride, rgba_stride, s.pics[0], mbs, words)
        assert host is not None and err == NO_ERROR
        _compare_packed(host, dc, dw, 0)
        for c, pos in zip(host[0], _chunk_positions(host[0])):
            if len(pos) < 4:
                continue
            raster = [y * mb_w + x for x, y in pos]
            consecutive = raster == list(range(raster[0], raster[0] + 4))
            wraps = pos[0][1] != pos[3][1]
            if bool(int(c[4]) & K_CRUN):
                assert consecutive                          # a run's tiles are consecutive, in one row or across a row end
                wrapped_runs += wraps
            if consecutive and s.picture_type != desc.PIC_I:
                # predicted macroblocks write all their blocks: every such chunk is a run (an intra macroblock with an invalid block is not)
                intra = [int(c[8 + 6 * m]) & 1 for m in range(4)]
                cbp = [(int(c[8 + 6 * m]) >> 8) & 0x3f for m in range(4)]
                assert bool(int(c[4]) & K_CRUN) == all((not i) or b == 0x3f for i, b in zip(intra, cbp))
    assert wrapped_runs > 0
    # ... and the pictures reconstruct like the oracle's, planes and RGBA, through both packers and both kernel instances
    for device_pack in (0, 1):
        emu.set_device_pack(device_pack)
        try:
            for tile in (1, 2):
                emu.set_tile_policy(tile)
                ref, dut = oracle.OracleStore(w, h), emu.EmuStore(w, h)
                for i, s in enumerate(seq):
                    mbs, words = desc.to_sparse(s.mbs, s.coefs)
                    ref.submit(s.pics, s.mbs, s.coefs)
                    dut.submit_sparse(s.pics[0], mbs, words)
                    for slot in range(3):
                        for a, b in zip(ref.read_planes(0, slot), dut.read_planes(0, slot)):
                            assert np.array_equal(a, b), (device_pack, tile, i, slot)
                    if rgba:
                        assert np.array_equal(ref.read_rgba(0, s.cur), dut.read_rgba(0, s.cur)), (device_pack, tile, i)
        finally:
            emu.set_tile_policy(0)
            emu.set_device_pack(0)
