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


@pytest.mark.parametrize("w,h,n,profile", [(96, 80, 7, "typical"), (160, 128, 5, "dense"), (64, 48, 10, "typical"), (37, 23, 4, "typical")])
def test_written_stream_decodes_to_the_descriptors_it_was_written_from(oracle, w, h, n, profile):
    seq = synth.generate_sequence(w, h, n, seed=0x77 + w, profile=profile)
    es = mpeg1_writer.write_sequence(w, h, seq)
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
