"""-m gpu: the reference's own golden tests, run through the PRODUCT: host parser (libmpeghost) ->
descriptors -> C ABI (libmpeghip) -> HIP kernels on the MI355X -> planes / samples back on the host.
TestVideoGolden (mpeg_test.go:205-231) and TestAudioGolden (mpeg_test.go:166-201)."""
import ctypes as C

import numpy as np
import pytest

import hostlib

pytestmark = pytest.mark.gpu

VIDEO_HASH = 0xea6d7fcb1340ba3f
TESTMPG_VIDEO_HASH = 0xd00818edcafdc702


@pytest.fixture(scope="module")
def device():
    d = hostlib.host().mpeghost_device_create(0)
    assert d, hostlib.host().mpeghost_last_error()
    yield d
    hostlib.host().mpeghost_device_destroy(d)


def test_video_golden_hash_on_gpu(oracle, golden_dir, device):
    dec = hostlib.HostVideo((golden_dir / "test.mpeg1video").read_bytes(), device=device)
    assert (hostlib.host().mpeghost_video_width(dec.h), hostlib.host().mpeghost_video_height(dec.h)) == (160, 120)
    h, n = oracle.FNV_OFFSET, 0
    while True:
        f = dec.decode()
        if f is None:
            break
        for p in hostlib.frame_planes(f):
            h = oracle.fnv1a64(p, h)
        n += 1
    st = dec.stats()
    dec.close()
    assert (h, n) == (VIDEO_HASH, 260)
    assert st["invalid_blocks"] == 53 and st["raw_macroblocks"] > 0


@pytest.mark.parametrize("pack_from", [0, 1], ids=["host_packed", "device_packed"])
@pytest.mark.parametrize("mirror", [True, False], ids=["mirror", "read_back"])
def test_video_golden_hash_through_either_hand_over_and_either_way_back(oracle, golden_dir, device, pack_from, mirror):
    """a lone decoder's pictures validated and packed on the host or — every one of them, also these 80-macroblock ones — by the
    device (a device-packed stage of one picture: mpeghost_video_set_device_pack_from), its frames out of the store's host mirror
    or read back: the reference's hash of the damaged golden stream all four ways"""
    dec = hostlib.HostVideo((golden_dir / "test.mpeg1video").read_bytes(), device=device)
    dec.set_device_pack_from(pack_from)
    dec.set_host_mirror(mirror)
    h, n = oracle.FNV_OFFSET, 0
    while True:
        f = dec.decode()
        if f is None:
            break
        for p in hostlib.frame_planes(f):
            h = oracle.fnv1a64(p, h)
        n += 1
    dec.close()
    assert (h, n) == (VIDEO_HASH, 260)


def test_frame_rgba_on_gpu_matches_oracle(oracle, golden_dir, device):
    data = (golden_dir / "test.mpeg1video").read_bytes()
    ref, dut = oracle.VideoDecoder(data), hostlib.HostVideo(data, device=device)
    for i in range(12):
        a, b = ref.decode(), dut.decode()
        for pa, pb in zip(oracle.frame_planes(a), hostlib.frame_planes(b)):
            assert np.array_equal(pa, pb), "frame %d" % i
        want = np.empty((120, 160, 4), np.uint8)
        oracle.lib().orc_ycbcr_to_rgba(C.byref(a), want.ctypes.data)
        assert np.array_equal(want, dut.rgba(160, 120)), "Frame.RGBA of frame %d" % i
    ref.close()
    dut.close()


@pytest.mark.parametrize("fma,want", [(0, 0xf1b76cdf8e6cdea5), (1, 0x50f3ab75f5fb0fb5)])
def test_audio_golden_hash_on_gpu(oracle, golden_dir, device, fma, want):
    dec = hostlib.HostAudio((golden_dir / "test.mp2").read_bytes(), device=device, fma=fma)
    h, n = oracle.FNV_OFFSET, 0
    while True:
        s = dec.decode()
        if s is None:
            break
        h = oracle.fnv1a64(s, h)
        n += 1
    dec.close()
    assert (h, n) == (want, 355)


def test_program_stream_facade_on_gpu(oracle, golden_dir, device):
    """mpeg.New + DecodeVideo / DecodeAudio on test.mpg (TestMpeg / BenchmarkDecodeVideo's input, BASELINE config 1)."""
    L = hostlib.host()
    ps = (golden_dir / "test.mpg").read_bytes()
    m = L.mpeghost_mpeg_open(device, ps, len(ps))
    assert m, L.mpeghost_last_error()
    info = (C.c_int * 6)()
    L.mpeghost_mpeg_info(m, C.byref(info))
    assert list(info) == [1, 1, 160, 120, 44100, 1]            # mpeg_test.go:300-330
    assert L.mpeghost_mpeg_framerate(m) == 30.0
    # video only, then audio only (the reference's benchmarks disable the other stream the same way)
    L.mpeghost_mpeg_set_enabled(m, 1, 0)
    h, n, f = oracle.FNV_OFFSET, 0, hostlib.HostFrame()
    while L.mpeghost_mpeg_decode_video(m, C.byref(f)) == 1:
        for p in hostlib.frame_planes(f):
            h = oracle.fnv1a64(p, h)
        n += 1
    assert (h, n) == (TESTMPG_VIDEO_HASH, 278)
    assert L.mpeghost_mpeg_has_ended(m) == 1
    L.mpeghost_mpeg_close(m)

    m = L.mpeghost_mpeg_open(device, ps, len(ps))
    L.mpeghost_mpeg_set_enabled(m, 0, 1)
    h, n, t = oracle.FNV_OFFSET, 0, C.c_double()
    while True:
        p = L.mpeghost_mpeg_decode_audio(m, C.byref(t))
        if not p:
            break
        h = oracle.fnv1a64(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(2304,)), h)
        n += 1
    assert (h, n) == (0xf1b76cdf8e6cdea5, 355)
    L.mpeghost_mpeg_close(m)


def test_seek_host_logic_runs_the_same_over_either_backend_no_parity_claim(emu, golden_dir, device):
    """MPEG.Seek / SeekFrame (mpeg.go:460-576) are host logic: frame times, the callback count of a seek (mpeg_test.go:
    442-461: exactly one) and the audio / video clocks must come out the same whichever backend reconstructs the frames.
    (NOT a parity test of the frames: those are compared with the ORACLE's in
    tests/test_gpu_parity_holes.py::test_exact_seek_on_gpu_returns_pictures_of_the_oracles_linear_decode.)"""
    ps = (golden_dir / "test.mpg").read_bytes()
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    gpu, cpu = hostlib.HostMpeg(ps, device=device), hostlib.HostMpeg(ps, window=win)
    # same call sequence on both: the seek estimator starts from the demuxer's last decoded PTS (demux.go:244)
    assert abs(gpu.duration - 9.233333) < 1e-3 and cpu.duration == gpu.duration   # mpeg_test.go:104
    for t, exact in ((3.0, True), (1.0, False), (6.25, True), (100.0, True), (0.0, True)):
        fg, fc = gpu.seek_frame(t, exact), cpu.seek_frame(t, exact)
        assert fg is not None and fc is not None and fg.time == fc.time
    for m in (gpu, cpu):
        m.count_callbacks()
        assert m.seek(3.001, True)
    assert gpu.callback_counts() == cpu.callback_counts() and gpu.callback_counts()[0] == 1   # mpeg_test.go:442-461
    assert gpu.time == cpu.time and gpu.audio_time == cpu.audio_time and abs(gpu.audio_time - gpu.time) <= 0.5
    gpu.close()
    cpu.close()


def test_video_batch_on_gpu(oracle, golden_dir, device):
    """mpeg::VideoBatch over the HIP store: staggered, mixed streams reconstructed by one device call per tick,
    each bit-identical to its golden hash (tests/test_host_batch.py runs the same on the lane emulator)."""
    from test_host_batch import TESTMPG_VIDEO_HASH as CLEAN, VIDEO_HASH as DAMAGED, run_batch
    es = (golden_dir / "test.mpeg1video").read_bytes()
    clean = oracle.ps_extract((golden_dir / "test.mpg").read_bytes(), 0xE0)[0]
    streams = [es, clean, es, es, clean, es, es, clean]
    h, n, c = run_batch(oracle, streams, [0, 0, 1, 5, 9, 2, 2, 3], device=device)
    assert h == [DAMAGED if s is es else CLEAN for s in streams]
    assert n == [260 if s is es else 278 for s in streams]
    assert c["device_submits"] < c["queued_pictures"] / 4
    # the same with the streams parsed on 4 host threads (VideoBatch::SetThreads)
    h4, n4, c4 = run_batch(oracle, streams, [0, 0, 1, 5, 9, 2, 2, 3], device=device, threads=4)
    assert (h4, n4) == (h, n) and c4["device_submits"] <= c["device_submits"]


def test_written_1080p_stream_through_the_parser_and_the_gpu(oracle, device):
    """A 1920x1080 stream (tests/mpeg1_writer.py over the seeded descriptor generator: I P B B, the typical mix) through
    the PRODUCT PATH — bitstream parser -> descriptors -> C ABI -> kernels — frame by frame against the oracle's
    reconstruction of the descriptors the stream was written from; then six copies of it through mpeg::VideoBatch
    with three parse threads (staged submits)."""
    import mpeg1_writer
    from mpeg_amd import synth
    from test_written_streams import decode_all, expected_frames
    w, h = 1920, 1080
    seq = synth.generate_sequence(w, h, 4, seed=0x1080)
    es = mpeg1_writer.write_sequence(w, h, seq)
    want = expected_frames(oracle, w, h, seq)
    assert len(want) == 3                      # I, B, B (the P picture is still held when the stream ends on a B)
    for pack_from in (None, 0):                # the default (a 1080p picture is packed by the device), and packed on the host
        dut = hostlib.HostVideo(es, device=device)
        if pack_from is not None:
            dut.set_device_pack_from(pack_from)
        got = decode_all(dut, hostlib.frame_planes)
        st = dut.stats()
        dut.close()
        assert len(got) == len(want) and st["pictures"] == 4 and st["invalid_blocks"] == 0 and st["range_skips"] == 0
        for i, (a, b) in enumerate(zip(want, got)):
            for pa, pb in zip(a, b):
                assert np.array_equal(pa, pb), "frame %d (device_pack_from %s)" % (i, pack_from)
    b = hostlib.HostBatch(6, device=device, threads=3)
    for _ in range(6):
        b.add_stream(es)
    for i in range(len(want)):
        assert b.decode_all() == 6
        for k in range(6):
            for pa, pb in zip(want[i], hostlib.frame_planes(b.frame(k))):
                assert np.array_equal(pa, pb), "batch: stream %d frame %d" % (k, i)
    assert b.decode_all() == 0
    b.close()


def test_audio_batch_on_gpu(oracle, device):
    """mpeg::AudioBatch over the HIP store: staggered MP2 streams, one synthesis call per tick, every stream's
    samples hash to the reference's golden value (tests/test_host_batch.py runs the same on the lane emulator)."""
    from test_host_batch import AUDIO_HASH, run_audio_batch
    h, cnt, calls = run_audio_batch(oracle, 6, [0, 0, 3, 7, 40, 1], device=device)
    assert h == [AUDIO_HASH] * 6 and cnt == [355] * 6 and calls == 355 + 40
    h, cnt, _ = run_audio_batch(oracle, 3, [0, 2, 2], fmt=3, device=device)   # S16
    assert len(set(h)) == 1 and cnt == [355] * 3
    # the streams' frames of a tick parsed on four host threads (AudioBatch::SetThreads): the same samples, the same calls
    h, cnt, calls = run_audio_batch(oracle, 9, [0, 0, 3, 7, 40, 1, 0, 5, 2], device=device, threads=4)
    assert h == [AUDIO_HASH] * 9 and cnt == [355] * 9 and calls == 355 + 40
