"""The product's host library (mpeg_amd/host: Buffer, Video/Audio parse -> descriptors) checked on the CPU:
its parser drives a TEST-ONLY backend built from the kernels' lane functions, and the decoded output must
reproduce the reference's golden hashes (mpeg_test.go:193-197, :227).  The same path with the real HIP
backend runs under -m gpu in test_gpu_golden.py."""
import numpy as np
import pytest

import hostlib

VIDEO_HASH = 0xea6d7fcb1340ba3f
TESTMPG_VIDEO_HASH = 0xd00818edcafdc702


def video_hash(oracle, dec):
    h, n = oracle.FNV_OFFSET, 0
    while True:
        f = dec.decode()
        if f is None:
            break
        for p in hostlib.frame_planes(f):
            h = oracle.fnv1a64(p, h)
        n += 1
    return h, n


@pytest.mark.parametrize("flavour", [0, 1])  # 0: wave-chunk kernel lane code, 1: fused kernel lane code
def test_damaged_golden_stream(oracle, golden_dir, flavour):
    dec = hostlib.HostVideo((golden_dir / "test.mpeg1video").read_bytes(), emu_flavour=flavour)
    h, n = video_hash(oracle, dec)
    st = dec.stats()
    dec.close()
    assert (h, n) == (VIDEO_HASH, 260)
    assert st["invalid_blocks"] == 53 and st["range_skips"] == 0
    assert st["raw_macroblocks"] > 0       # stale blockData really occurs in this stream and goes through the snapshot path
    assert st["pictures"] == 261


def test_clean_stream_from_program_stream(oracle, golden_dir):
    es, _ = oracle.ps_extract((golden_dir / "test.mpg").read_bytes(), 0xE0)
    dec = hostlib.HostVideo(es, emu_flavour=0)
    h, n = video_hash(oracle, dec)
    st = dec.stats()
    dec.close()
    assert (h, n) == (TESTMPG_VIDEO_HASH, 278)
    assert st["raw_macroblocks"] == 0 and st["invalid_blocks"] == 0 and st["duplicate_splits"] == 0


def test_frames_match_oracle_frame_by_frame(oracle, golden_dir):
    data = (golden_dir / "test.mpeg1video").read_bytes()
    ref, dut = oracle.VideoDecoder(data), hostlib.HostVideo(data, emu_flavour=0)
    for i in range(40):
        a, b = ref.decode(), dut.decode()
        assert (a is None) == (b is None)
        for pa, pb in zip(oracle.frame_planes(a), hostlib.frame_planes(b)):
            assert np.array_equal(pa, pb), "frame %d" % i
        assert abs(a.time - b.time) < 1e-12
    ref.close()
    dut.close()


@pytest.mark.parametrize("fma,want", [(0, 0xf1b76cdf8e6cdea5), (1, 0x50f3ab75f5fb0fb5)])
def test_audio_golden(oracle, emu, golden_dir, fma, want):
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    dec = hostlib.HostAudio((golden_dir / "test.mp2").read_bytes(), fma=fma, window=win)
    h, n = oracle.FNV_OFFSET, 0
    while True:
        s = dec.decode()
        if s is None:
            break
        h = oracle.fnv1a64(s, h)
        n += 1
    assert (dec.samplerate, dec.channels) == (44100, 1)
    dec.close()
    assert (h, n) == (want, 355)
