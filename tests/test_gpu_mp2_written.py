"""-m gpu: the written MP2 streams of tests/mp2_writer.py (stereo, joint stereo with all four bounds, dual channel, mono;
allocation tables A-D; 32 / 44.1 / 48 kHz; CRC and padding) through the product — host parser -> C ABI -> audio_kernel on the
MI355X — against the oracle's decode of the same bytes, both arithmetic modes, single decoders and an AudioBatch."""
import zlib

import numpy as np
import pytest

import hostlib
import mp2_writer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    d = hostlib.host().mpeghost_device_create(0)
    assert d, hostlib.host().mpeghost_last_error()
    yield d
    hostlib.host().mpeghost_device_destroy(d)


@pytest.mark.parametrize("fma", [0, 1], ids=["no_fma", "fma_window"])
@pytest.mark.parametrize("case", mp2_writer.CASES, ids=[c[0] for c in mp2_writer.CASES])
def test_written_mp2_streams_on_the_gpu(oracle, device, case, fma):
    name, mode, br, sr, bound, crc = case
    n_frames = 8
    es, frames = mp2_writer.write_stream(n_frames, mode, br, sr, bound, crc, seed=zlib.crc32(name.encode()) & 0xffff)
    ref = oracle.AudioDecoder(es, fma)
    dut = hostlib.HostAudio(es, device=device, fma=fma)
    for i in range(n_frames):
        pcm, samples = ref.decode(True)
        assert np.array_equal(samples, mp2_writer.expected_samples(frames[i])), "%s frame %d: sub-band samples" % (name, i)
        got = dut.decode()
        assert got is not None and np.array_equal(pcm.view(np.uint32), got.view(np.uint32)), "%s frame %d: PCM differs" % (name, i)
    assert ref.decode() is None and dut.decode() is None
    ref.close()
    dut.close()


def test_written_streams_in_one_audio_batch_on_the_gpu(oracle, device):
    """All fifteen written streams side by side in ONE mpeg::AudioBatch: one synthesis launch per tick for streams of every mode."""
    cases = mp2_writer.CASES
    streams = [mp2_writer.write_stream(6, c[1], c[2], c[3], c[4], c[5], seed=300 + i)[0] for i, c in enumerate(cases)]
    batch = hostlib.HostAudioBatch(len(streams), device=device, fmt=0, fma=0)
    for es in streams:
        batch.add_stream(es)
    refs = [oracle.AudioDecoder(es, 0) for es in streams]
    for _ in range(6):
        assert batch.decode_all() == len(streams)
        for k, ref in enumerate(refs):
            want = ref.decode()
            got = np.asarray(batch.samples(k)).reshape(-1)[:2304]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "stream %d (%s)" % (k, cases[k][0])
    assert batch.device_calls == 6
    batch.close()
    for r in refs:
        r.close()
