"""The sparse hand-over (include/mpeghip.h: mpeghip_video_stage_put_sparse — the parser's (position, level) pairs straight into
device entries) on the CPU: the library's sparse packer + the kernel's lane functions (lane emulator) against the oracle and
against the unit form of the same pictures; the validator's refusals.  The GPU twin is tests/test_gpu_sparse.py."""
import numpy as np
import pytest

from mpeg_amd import desc, synth
from parity import assert_planes_equal


def _run_both(oracle, emu, w, h, seq, tile):
    emu.set_tile_policy(tile)
    try:
        ref, a, b = oracle.OracleStore(w, h), emu.EmuStore(w, h), emu.EmuStore(w, h)
        for s in seq:
            ref.submit(s.pics, s.mbs, s.coefs)
            a.submit(s.pics, s.mbs, s.coefs)
            mbs, words = desc.to_sparse(s.mbs, s.coefs)
            assert b.submit_sparse(s.pics[0], mbs, words) == 0
            for slot in range(3):
                assert_planes_equal(ref.read_planes(0, slot), b.read_planes(0, slot), "sparse vs oracle, slot %d" % slot)
                assert_planes_equal(a.read_planes(0, slot), b.read_planes(0, slot), "sparse vs units, slot %d" % slot)
            if s.pics["flags"][0] & desc.PIC_RGBA:
                assert np.array_equal(ref.read_rgba(0, s.cur), b.read_rgba(0, s.cur))
    finally:
        emu.set_tile_policy(0)


@pytest.mark.parametrize("tile", [1, 2], ids=["int16", "int32"])
@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 6, "typical", 0.0, False),
    (352, 240, 4, "typical", 0.15, False),   # snapshot blocks
    (352, 240, 3, "dense", 0.0, False),      # every block beyond 32 levels: units are built from the pairs
    (160, 120, 5, "typical", 0.05, True),
    (50, 35, 3, "typical", 0.0, True),
])
def test_sparse_pictures_match_the_oracle_and_the_unit_form(oracle, emu, w, h, n, profile, raw, rgba, tile):
    _run_both(oracle, emu, w, h, synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba), tile)


def test_a_coded_zero_level_dequantises_like_the_reference(oracle, emu):
    """video.go:719-736: a coded level of 0 becomes +-1 after the oddification — units (0 = absent) cannot say that (the
    parser sends such blocks as snapshots), a pair can.  Checked against the oracle's own dequantiser through a snapshot
    of the same block."""
    w, h = 48, 32
    g = desc.geometry(w, h)
    seq = synth.generate_sequence(w, h, 2, seed=5)
    s = seq[1]   # a P picture on top of the I picture
    ref, dut = oracle.OracleStore(w, h), emu.EmuStore(w, h)
    ref.submit(seq[0].pics, seq[0].mbs, seq[0].coefs)
    dut.submit(seq[0].pics, seq[0].mbs, seq[0].coefs)
    k = next(i for i, m in enumerate(s.mbs) if m["cbp"] and not (m["flags"] & (desc.MB_INTRA | desc.MB_COEF_RAW)))
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    words = list(words)
    at = int(mbs[k]["coef_off"])
    n = words[at]
    used = {(x >> 2) & 63 for x in words[at + 1:at + 1 + n]}
    free = [p for p in range(64) if p not in used][:3]
    words[at] = n + len(free)
    words[at + 1 + n:at + 1 + n] = [p << 2 for p in free]          # three coded zeros
    for j in range(k + 1, len(mbs)):
        mbs[j]["coef_off"] += len(free)
    assert dut.submit_sparse(s.pics[0], mbs, np.array(words, np.uint32)) == 0
    # the oracle's view: the same macroblock as a snapshot of what the reference's loop leaves in blockData
    def blockdata(level, qs, qm, pm):   # video.go:719-744, non-intra
        l = level << 1
        l += -1 if l < 0 else 1         # (0 counts as "else": +1)
        l = (l * qs * qm) >> 4
        if (l & 1) == 0:
            l -= 1 if l > 0 else -1
        return max(-2048, min(2047, l)) * pm

    qs = int(s.mbs[k]["qscale"])
    units = s.coefs.view(np.int16).reshape(-1, 64)
    nb = bin(int(s.mbs[k]["cbp"])).count("1")
    snap = np.zeros((nb, 64), np.int32)
    premult = np.asarray(synth.PREMULT).reshape(-1)
    for b in range(nb):
        u = units[int(s.mbs[k]["coef_off"]) + b]
        for pos in range(64):
            if u[pos] or (b == 0 and pos in free):
                nat = (pos & 7) * 8 + (pos >> 3)
                snap[b, pos] = blockdata(int(u[pos]), qs, int(np.asarray(synth.NON_INTRA_Q).reshape(-1)[nat]), int(premult[nat]))
    m2 = s.mbs.copy()
    other = s.coefs.view(np.uint8).reshape(-1)
    m2[k]["flags"] |= desc.MB_COEF_RAW
    m2[k]["coef_off"] = len(other) // 128
    ref.submit(s.pics, m2, np.concatenate([other, snap.reshape(-1).view(np.uint8)]))
    for slot in range(3):
        assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(0, slot), "coded zeros, slot %d" % slot)
    assert g["mb_count"] == len(mbs)


@pytest.mark.parametrize("damage,why", [
    (lambda w, at: w.__setitem__(at, 65), "a count beyond 64"),
    (lambda w, at: w.__setitem__(at + 1, w[at + 1] | 0x0100), "stray bits in a pair"),
    (lambda w, at: w.__delitem__(slice(at + 3, None)), "a block beyond the picture's words"),
])
def test_malformed_block_data_is_refused(emu, damage, why):
    w, h = 48, 32
    s = synth.generate_sequence(w, h, 1, seed=8, profile="dense")[0]
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    words = list(int(x) for x in words)
    damage(words, int(mbs[2]["coef_off"]))
    assert emu.EmuStore(w, h).submit_sparse(s.pics[0], mbs, np.array(words, np.uint32)) == -2, why


def test_an_intra_block_needs_its_dc_first(emu):
    w, h = 48, 32
    s = synth.generate_sequence(w, h, 1, seed=8)[0]
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    k = next(i for i, m in enumerate(mbs) if (m["flags"] & desc.MB_INTRA) and not (m["flags"] & desc.MB_COEF_RAW) and m["cbp"])
    words = words.copy()
    words[int(mbs[k]["coef_off"]) + 1] |= 5 << 2      # the first pair is no longer position 0
    assert emu.EmuStore(w, h).submit_sparse(s.pics[0], mbs, words) == -2


def test_to_sparse_is_what_the_header_describes():
    s = synth.generate_sequence(48, 32, 1, seed=8)[0]
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    units = s.coefs.view(np.int16).reshape(-1, 64)
    k = next(i for i, m in enumerate(mbs) if m["cbp"] and not (m["flags"] & desc.MB_COEF_RAW))
    at, u = int(mbs[k]["coef_off"]), units[int(s.mbs[k]["coef_off"])]
    n = int(words[at])
    got = {int((x >> 2) & 63): np.int16(np.uint16(x >> 16)) for x in words[at + 1:at + 1 + n]}
    want = {int(p): u[p] for p in np.nonzero(u)[0]}
    if s.mbs[k]["flags"] & desc.MB_INTRA:
        want.setdefault(0, np.int16(0))
    assert got == want


def test_mutated_sparse_words_are_refused_or_reconstructed_never_worse(oracle, emu):
    """Bit flips in count and pair words: the packer either refuses the picture (-2) or packs it — then whatever it accepted
    is a well-formed picture (counts within the buffer, no stray bits), which the lane functions reconstruct with all of their
    range checks on (MPG_EMU_CHECKS) and which the oracle reconstructs to the same bytes from the equivalent units, unless a
    position was named twice (unspecified by the ABI: then only the absence of a crash is checked)."""
    import random
    w, h = 64, 48
    s = synth.generate_sequence(w, h, 2, seed=17, profile="typical")[1]
    first = synth.generate_sequence(w, h, 2, seed=17, profile="typical")[0]
    mbs, words0 = desc.to_sparse(s.mbs, s.coefs)
    refused = accepted = compared = 0
    for seed in range(150):
        rng = random.Random(seed)
        words = words0.copy()
        for _ in range(rng.randrange(1, 4)):
            words[rng.randrange(len(words))] ^= np.uint32(1 << rng.randrange(32))
        dut = emu.EmuStore(w, h)
        dut.submit(first.pics, first.mbs, first.coefs)
        rc = dut.submit_sparse(s.pics[0], mbs, words)
        assert rc in (0, -2)
        if rc == -2:
            refused += 1
            continue
        accepted += 1
        # the same picture as units (possible when no snapshot data was hit, no level is a coded zero and no position repeats)
        units = np.zeros((int(sum(bin(int(c)).count("1") for c in mbs["cbp"])), 64), np.int16)
        m2, ok, u = s.mbs.copy(), True, 0
        for k in range(len(mbs)):
            at = int(mbs[k]["coef_off"])
            if mbs[k]["flags"] & desc.MB_COEF_RAW:
                ok = False
                break
            m2[k]["coef_off"] = u
            for _ in range(bin(int(mbs[k]["cbp"])).count("1")):
                n = int(words[at])
                pairs = words[at + 1:at + 1 + n]
                pos = (pairs >> 2) & 63
                lev = (pairs >> 16).astype(np.uint16).view(np.int16)
                intra = bool(mbs[k]["flags"] & desc.MB_INTRA)
                if len(set(pos.tolist())) != n or (lev[1 if intra else 0:] == 0).any():
                    ok = False
                units[u, pos] = lev
                u += 1
                at += 1 + n
        if not ok:
            continue
        ref = oracle.OracleStore(w, h)
        ref.submit(first.pics, first.mbs, first.coefs)
        ref.submit(s.pics, m2, units.reshape(-1).view(np.uint8))
        for slot in range(3):
            assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(0, slot), "seed %d slot %d" % (seed, slot))
        compared += 1
    assert refused >= 20 and accepted >= 20 and compared >= 10, (refused, accepted, compared)


@pytest.mark.parametrize("w,h,profile,raw", [(352, 240, "typical", 0.2), (96, 64, "dense", 0.0), (50, 35, "typical", 0.5)])
def test_the_vectorised_to_sparse_equals_the_block_by_block_one(w, h, profile, raw):
    for keep in (True, False):
        for s in synth.generate_sequence(w, h, 3, profile=profile, raw_fraction=raw, seed=w):
            a, b = desc.to_sparse(s.mbs, s.coefs, keep), desc.to_sparse_loop(s.mbs, s.coefs, keep)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
