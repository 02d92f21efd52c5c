"""Written MP2 streams (tests/mp2_writer.py) through the two frame parsers — the oracle's (restating audio.go:184-490) and the
product's (mpeg_amd/host/audio.cpp) — in every mode the one golden file does not reach: stereo, joint stereo with all four
bounds, dual channel, mono; allocation tables 3-B.2a-d; 32 / 44.1 / 48 kHz; with and without CRC word and padding slot.
Three legs per stream: what the writer put in (its own restatement of the requantisation, from the standard), the oracle's
sub-band samples and PCM, the product's PCM through the lane emulator (the GPU twin: tests/test_gpu_mp2_written.py)."""
import zlib

import numpy as np
import pytest

import hostlib
import mp2_writer


@pytest.mark.parametrize("case", mp2_writer.CASES, ids=[c[0] for c in mp2_writer.CASES])
def test_written_mp2_streams_through_both_parsers(oracle, emu, case):
    name, mode, br, sr, bound, crc = case
    n_frames = 6
    es, frames = mp2_writer.write_stream(n_frames, mode, br, sr, bound, crc, seed=zlib.crc32(name.encode()) & 0xffff)
    sblimit, _, table = mp2_writer.table_of(mode, br, sr)
    assert table == name.split("_")[3] and frames[0]["sblimit"] == sblimit
    assert any(f["q"].any() for f in frames), "the frames carry no allocation at all"
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    for fma in (0, 1):
        ref = oracle.AudioDecoder(es, fma)
        dut = hostlib.HostAudio(es, fma=fma, window=win)
        assert ref.samplerate == dut.samplerate == mp2_writer.SAMPLERATE[sr]
        assert ref.channels == dut.channels == (1 if mode == mp2_writer.MODE_MONO else (0 if mode == mp2_writer.MODE_DUAL else 2))
        for i in range(n_frames):
            r = ref.decode(True)
            assert r is not None, "oracle: frame %d missing" % i
            pcm, samples = r
            want = mp2_writer.expected_samples(frames[i])
            assert np.array_equal(samples, want), "%s frame %d: the oracle's sub-band samples differ from what was written" % (name, i)
            got = dut.decode()
            assert got is not None, "product: frame %d missing" % i
            assert np.array_equal(pcm.view(np.uint32), got.view(np.uint32)), "%s frame %d fma %d: PCM differs" % (name, i, fma)
        assert ref.decode() is None and dut.decode() is None
        ref.close()
        dut.close()


def test_the_cases_cover_every_mode_table_and_rate():
    modes = {c[1] for c in mp2_writer.CASES}
    tables = {mp2_writer.table_of(c[1], c[2], c[3])[2] for c in mp2_writer.CASES}
    rates = {c[3] for c in mp2_writer.CASES}
    bounds = {c[4] for c in mp2_writer.CASES if c[1] == mp2_writer.MODE_JOINT}
    assert modes == {0, 1, 2, 3} and tables == set("ABCD") and rates == {0, 1, 2} and bounds == {0, 1, 2, 3}
    assert {c[5] for c in mp2_writer.CASES} == {True, False} and len(mp2_writer.CASES) >= 12


def test_written_streams_through_the_audio_batch(oracle, emu):
    """Five written streams of different modes side by side in one mpeg::AudioBatch (one synthesis call per tick)."""
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    cases = [mp2_writer.CASES[i] for i in (0, 5, 9, 12, 8)]
    streams = [mp2_writer.write_stream(5, c[1], c[2], c[3], c[4], c[5], seed=77 + i)[0] for i, c in enumerate(cases)]
    batch = hostlib.HostAudioBatch(len(streams), fmt=0, fma=0, window=win)
    for es in streams:
        batch.add_stream(es)
    refs = [oracle.AudioDecoder(es, 0) for es in streams]
    for _ in range(5):
        assert batch.decode_all() == len(streams)
        for k, ref in enumerate(refs):
            want = ref.decode()
            got = batch.samples(k)
            got = got[0] if isinstance(got, tuple) else got
            assert np.array_equal(np.asarray(got).view(np.uint32).reshape(-1)[:2304], want.view(np.uint32))
    batch.close()
    for r in refs:
        r.close()
