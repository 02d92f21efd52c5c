"""Synthetic descriptor sequences written out as MPEG-1 elementary streams (tests/mpeg1_writer.py) and decoded
again — by the oracle's parser and by the product's parser — so that pictures of ANY size reach the product
through its bitstream parser, not only through the descriptor ABI.  Three results must agree picture by picture:

    oracle reconstruction of the ORIGINAL descriptors            (what the writer meant)
    oracle parser + reconstruction of the written stream         (the reference's reading of the stream)
    product parser -> descriptors -> kernels (lane emulator here, the GPU in test_gpu_golden.py)
"""
import numpy as np
import pytest

import hostlib
import mpeg1_writer
from mpeg_amd import desc, synth


def display_order(types):
    """Indices of the decode-order pictures in the order Video.Decode returns them (video.go:209-268)."""
    out, held = [], None
    for i, t in enumerate(types):
        if t == desc.PIC_B:
            out.append(i)
        else:
            if held is not None:
                out.append(held)
            held = i
    if held is not None and types[-1] != desc.PIC_B:
        out.append(held)  # flushed at the end of the stream only if the last picture was a reference (video.go:220-229)
    return out


def expected_frames(oracle, w, h, seq):
    """What every picture looks like right after it was decoded, from the descriptors (no bitstream involved)."""
    ref = oracle.OracleStore(w, h, 1)
    pictures = []
    for s in seq:
        ref.submit(s.pics, s.mbs, s.coefs)
        pictures.append(ref.read_planes(0, s.cur))
    ref.close()
    return [pictures[i] for i in display_order([s.picture_type for s in seq])]


def decode_all(dec, planes_of):
    frames = []
    while True:
        f = dec.decode()
        if f is None:
            return frames
        frames.append(planes_of(f))


@pytest.mark.parametrize("table", [True, False], ids=["table_b5", "escapes"])
@pytest.mark.parametrize("w,h,n,profile", [(96, 80, 7, "typical"), (160, 128, 5, "dense"), (64, 48, 10, "typical"), (37, 23, 4, "typical"),
                                           (176, 144, 7, "natural")])
def test_written_stream_decodes_to_the_descriptors_it_was_written_from(oracle, w, h, n, profile, table):
    """table_b5: coefficients as the run / level codes of Table B.5c-g (video.go:1306-1419) wherever the table has one — every
    code of the table occurs in these streams (checked below for the natural profile) —, escapes elsewhere; escapes: every
    coefficient as an escape code (the parser's other path)."""
    seq = synth.generate_sequence(w, h, n, seed=0x77 + w, profile=profile)
    _three_way(oracle, w, h, seq, table)


def _three_way(oracle, w, h, seq, table):
    n = len(seq)
    es = mpeg1_writer.write_sequence(w, h, seq, table=table)
    want = expected_frames(oracle, w, h, seq)
    ref = oracle.VideoDecoder(es)
    assert (ref.width, ref.height, ref.framerate) == (w, h, 30.0)
    got_ref = decode_all(ref, oracle.frame_planes)
    ref.close()
    assert len(got_ref) == len(want)
    for i, (a, b) in enumerate(zip(want, got_ref)):
        for pa, pb in zip(a, b):
            assert np.array_equal(pa, pb), "oracle parser, frame %d" % i
    dut = hostlib.HostVideo(es, emu_flavour=0)
    got = decode_all(dut, hostlib.frame_planes)
    dst = dut.stats()
    dut.close()
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(want, got)):
        for pa, pb in zip(a, b):
            assert np.array_equal(pa, pb), "product parser, frame %d" % i
    assert dst["invalid_blocks"] == 0 and dst["range_skips"] == 0 and dst["raw_macroblocks"] == 0
    assert dst["pictures"] == n


def test_every_code_of_table_b5_goes_through_both_parsers(oracle):
    """A stream in which every (run, |level|) symbol of Table B.5c-g occurs with both signs, as a block's first symbol and as
    a later one, in intra and in non-intra blocks (the `1s` / `11s` forms of (0, 1) included), through the oracle's parser
    (the reference's tree, video.go:1306-1419) and the product's two-level tables: both reconstruct what the descriptors say."""
    w, h = 176, 144
    seq = synth.generate_sequence(w, h, 2, seed=0xB5, profile="dense")      # I, P: every macroblock has six coded blocks
    symbols = sorted(mpeg1_writer.COEFF) + [0x0001]
    assert len(symbols) == 111          # Table B.5c-g: 109 run / level codes + the `1` that is (0, 1) or end_of_block
    used = set()
    for s in seq:
        units = s.coefs.view(np.int16).reshape(-1, 64)
        intra = np.repeat((s.mbs["flags"] & desc.MB_INTRA) != 0, [bin(int(c)).count("1") for c in s.mbs["cbp"]])
        assert len(units) >= 4 * len(symbols)
        for b in range(len(units)):
            sym = symbols[(b // 4) % len(symbols)]
            run, level = sym >> 8, sym & 0xff
            sign = -1 if b & 1 else 1
            first = 1 if intra[b] else 0
            scan = np.zeros(64, np.int16)
            if intra[b]:
                scan[0] = units[b][0]                                       # (keep the DC)
            at = first
            if b & 2:                                                       # the symbol as a LATER one: something in front of it
                scan[at] = 3
                at += 1
            if at + run > 63:
                continue
            scan[at + run] = sign * level
            if at + run + 1 < 64:
                scan[at + run + 1] = -sign                                  # and (0, 1) behind it: the `11s` form
            nat = np.zeros(64, np.int16)
            nat[mpeg1_writer.ZIGZAG] = scan
            units[b] = nat[synth.TO_COLMAJOR] if hasattr(synth, "TO_COLMAJOR_INV") else nat.reshape(8, 8).T.reshape(-1)
            used.add((sym, bool(intra[b]), bool(b & 2), sign))
    for sym in symbols:
        for it in (False, True):
            assert {(sym, it, later, sg) for later in (False, True) for sg in (1, -1)} <= used or (sym >> 8) >= 62, hex(sym)
    _three_way(oracle, w, h, seq, True)


ESCAPE = "000001"


def _escape(run, level):
    return ESCAPE + format(run, "06b") + format(level & 0xff, "08b")


@pytest.mark.parametrize("name,bits", [
    # the block's coefficients end beyond position 63 with the offending symbol FOLLOWED BY '10' (the parser's table answers
    # "last coefficient + end_of_block" in one probe: the reference finds the overflow first and never reads the '10')
    ("overflow_then_end_of_block", "10" + _escape(62, 1) + "110" + "10"),
    ("overflow_then_more", "10" + _escape(62, 1) + "110" + "11" + "0"),
    ("overflow_by_run", "10" + _escape(30, 2) + mpeg1_writer.COEFF[0x1f01] + "0" + "110" + "10"),   # (run 31, level 1): 1 + 30 + 1 + 31 = 63, full; the next overflows
    # ... the parser's two-symbols-per-probe table: the first symbol fits (position 62), the second overflows; and the first itself
    ("overflow_by_the_second_of_a_pair", "10" + _escape(60, 1) + "110" + "0110" + "10"),
    ("overflow_by_the_first_of_a_pair", "10" + _escape(61, 1) + "0110" + "110" + "10"),
    ("overflow_by_escape", "10" + _escape(40, 1) + _escape(40, -1) + "10"),
    # a dead end of the code tree: twelve zeros read as run 0 / level 0, and the sign bit is still consumed (buffer.go:352-376)
    ("dead_end_code", "10" + "000000000000" + "1" + "0110" + "10"),
    ("coded_zero_by_escape", "10" + _escape(3, 0) + "00000000" + "110" + "10"),
    ("exactly_the_last_position", "10" + _escape(62, -5) + "10"),
])
@pytest.mark.parametrize("intra", [False, True], ids=["non_intra", "intra"])
def test_damaged_blocks_read_like_the_reference(oracle, name, bits, intra):
    """One block of a P picture (a non-intra macroblock's, or an intra macroblock's behind its DC) replaced by a hand-written
    symbol sequence: blocks that run past position 63 (video.go:711-714: the block is dropped, blockData stays dirty, the stream
    position stays right behind the symbol that overflowed), dead-end codes, coded zeros.  What follows the damage is read
    from wherever that leaves the cursor — the product's parser must land exactly where the oracle's (the reference's
    one-bit-at-a-time reading) does, and produce the same pictures."""
    w, h = 96, 80
    seq = synth.generate_sequence(w, h, 4, seed=0xD0, profile="typical")
    p = seq[1]
    assert p.picture_type == desc.PIC_P
    is_intra = (p.mbs["flags"] & desc.MB_INTRA) != 0
    # in every macroblock row (= slice: the damage ends at the next start code) the first macroblock of the wanted kind with two
    # coded blocks or more: its first coded block is damaged, so the next block is read from wherever the damage leaves the cursor
    mb_w = (w + 15) // 16
    targets = {}
    for i in range(len(p.mbs)):
        if bin(int(p.mbs["cbp"][i])).count("1") >= 2 and bool(is_intra[i]) == intra:
            targets.setdefault(i // mb_w, (i, next(b for b in range(6) if int(p.mbs["cbp"][i]) & (0x20 >> b))))
    assert targets
    if intra:
        bits = bits[2:] if bits.startswith("10") else bits      # (an intra block's first symbol is behind the DC: no `1s` form)
        bits = "110" + bits                                       # run 0 / level 1 as a later symbol
    hook = lambda pic, mb, blk, it: bits if pic == 1 and (mb, blk) in targets.values() else None
    es = mpeg1_writer.write_sequence(w, h, seq, block_hook=hook)
    assert es != mpeg1_writer.write_sequence(w, h, seq)
    ref = oracle.VideoDecoder(es)
    want = decode_all(ref, oracle.frame_planes)
    ref.close()
    for sparse in (1, 0):                       # the hand-over form of pairs, and the form of 128-byte units
        hostlib.host().mpeghost_set_default_sparse(sparse)
        try:
            dut = hostlib.HostVideo(es, emu_flavour=0)
            got = decode_all(dut, hostlib.frame_planes)
            st = dut.stats()
            dut.close()
        finally:
            hostlib.host().mpeghost_set_default_sparse(1)
        assert len(got) == len(want)
        for i, (a, b) in enumerate(zip(want, got)):
            for pa, pb in zip(a, b):
                assert np.array_equal(pa, pb), "%s, frame %d (sparse %d)" % (name, i, sparse)
        if name.startswith("overflow"):
            assert st["invalid_blocks"] >= 1, name      # (the damage is what it says it is)
