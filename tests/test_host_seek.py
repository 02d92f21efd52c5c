"""Program-stream timing and seeking of the product's host library (mpeg_amd/host/demux.cpp, mpeg.cpp),
held to the numbers the reference's own tests pin (mpeg_test.go:87-133 TestDemuxStartTimeDuration,
:276-398 TestMpeg, :402-438 TestSeekAudioTime, :442-461 TestSeekVideoCallbackOnce).  CPU only: the
decoders run on the test-only lane emulator; the same paths run on the HIP backend in test_gpu_golden.py."""
import numpy as np
import pytest

import hostlib
from hostlib import PACKET_AUDIO_1, PACKET_VIDEO_1


@pytest.fixture(scope="module")
def test_mpg(golden_dir):
    return (golden_dir / "test.mpg").read_bytes()


@pytest.fixture(scope="module")
def window(emu):
    return (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)


def near(got, want, eps=1e-3):
    return abs(got - want) <= eps


def test_demux_start_time_and_duration(test_mpg):
    """mpeg_test.go:87-133: per packet type, lowest / highest PTS (packets are reordered), + the last frame."""
    video_start = audio_start = 0.810078
    video_dur, audio_dur, first_video_pts = 9.233333, 9.325711, 0.876744
    d = hostlib.HostDemux(test_mpg)  # video first
    assert near(d.start_time(PACKET_VIDEO_1), video_start) and near(d.duration(PACKET_VIDEO_1), video_dur)
    assert near(d.start_time(PACKET_AUDIO_1), audio_start) and near(d.duration(PACKET_AUDIO_1), audio_dur)
    d.close()
    d = hostlib.HostDemux(test_mpg)  # audio first: the cache is keyed by type
    assert near(d.start_time(PACKET_AUDIO_1), audio_start) and near(d.duration(PACKET_AUDIO_1), audio_dur)
    assert near(d.start_time(PACKET_VIDEO_1), video_start) and near(d.duration(PACKET_VIDEO_1), video_dur)
    d.close()
    d = hostlib.HostDemux(test_mpg)
    assert d.start_time(PACKET_VIDEO_1) < first_video_pts  # looked past the first packet
    # the queries leave the read position alone: the first packet is still the first packet
    typ, pts, data = d.decode()
    assert typ == PACKET_VIDEO_1 and near(pts, first_video_pts)
    d.close()


def test_demux_probe_and_packet_census(test_mpg):
    """143 video + 37 audio packets; Probe counts what really occurs (demux.go:158-198)."""
    d = hostlib.HostDemux(test_mpg)
    assert d.streams() == (1, 1)
    assert d.probe(5000 * 1024) and d.streams() == (1, 1)
    d.rewind()
    n = {PACKET_VIDEO_1: 0, PACKET_AUDIO_1: 0}
    while True:
        p = d.decode()
        if p is None:
            break
        n[p[0]] = n.get(p[0], 0) + 1
    assert n[PACKET_VIDEO_1] == 143 and n[PACKET_AUDIO_1] == 37
    d.close()


def test_demux_seek_lands_on_an_intra_packet_before_the_time(test_mpg):
    """demux.go:208-352.  A fresh demuxer per query: the estimator starts from the last decoded PTS and
    byte position, and (in the reference's arithmetic too) a zero-length first jump that meets a
    reordered, lower PTS zeroes the byte-rate estimate for all 32 retries."""
    for t in (0.0, 1.0, 3.0, 4.5, 8.0, 100.0):
        d = hostlib.HostDemux(test_mpg)
        anchor = d.decode()[1]  # demux.go:236-238: Seek measures from the FIRST packet's PTS, not the lowest one
        d.rewind()
        span = d.duration(PACKET_VIDEO_1)
        p = d.seek(t, PACKET_VIDEO_1, True)
        assert p is not None and p[0] == PACKET_VIDEO_1, t
        i = p[2].find(b"\x00\x00\x01\x00")
        assert i >= 0 and (p[2][i + 5] & 0x38) == 8          # starts an intra picture
        assert p[1] - anchor <= min(t, span) + 1e-6
        assert min(t, span) - (p[1] - anchor) < 3.0 + 1e-6    # and the last one before it: test.mpg has one every 3 s
        d.close()


def test_mpeg_facade_numbers(test_mpg, window):
    """mpeg_test.go:276-398."""
    m = hostlib.HostMpeg(test_mpg, window=window)
    assert m.probe(5000 * 1024) and m.has_headers()
    i = m.info()
    assert (i["video_streams"], i["audio_streams"], i["width"], i["height"]) == (1, 1, 160, 120)
    assert m.framerate == 30.0 and i["samplerate"] == 44100 and i["channels"] == 1
    assert near(m.duration, 9.233333)
    m.set_enabled(True, False)
    f = m.decode_video()
    assert f is not None and f.luma_bytes == 20480 and f.chroma_bytes == 20480 // 4
    m.set_enabled(False, True)
    assert m.decode_audio() is not None
    m.set_enabled(True, True)
    assert m.seek(1.0, False)
    f = m.seek_frame(1.0, True)
    assert f is not None and f.time >= 1.0 - 1e-9 and f.time < 1.0 + 1.0 / 30 + 1e-9
    f = m.seek_frame(100.0, True)  # past the end: clamps to the duration, returns the last frame
    assert f is not None and f.time >= m.duration - 1.0
    m.count_callbacks()
    m.decode(1.0)
    m.close()


def test_rewind_after_the_end_clears_has_ended(test_mpg, window):
    """mpeg.go:323-337: Rewind resets time AND hasEnded — `for !m.HasEnded() { m.Decode(dt) }` must run again."""
    m = hostlib.HostMpeg(test_mpg, window=window)
    m.count_callbacks()
    ticks = 0
    while not m.has_ended and ticks < 400:
        m.decode(0.1)
        ticks += 1
    assert m.has_ended and 90 <= ticks < 400
    first = m.callback_counts()
    assert first[0] >= 270 and first[1] >= 350
    m.rewind()
    assert not m.has_ended and m.time == 0.0
    ticks = 0
    while not m.has_ended and ticks < 400:
        m.decode(0.1)
        ticks += 1
    second = m.callback_counts()
    assert m.has_ended and second[0] - first[0] == first[0] and second[1] - first[1] == first[1]   # the whole file once more
    m.close()


def test_loop_enabled_and_audio_stream_accessors(test_mpg, window):
    """mpeg.go:170-279, 338-345: getters mirror the setters; SetAudioStream takes 0..3 and re-selects the packet type."""
    m = hostlib.HostMpeg(test_mpg, window=window)
    assert m.get_enabled() == (True, True) and not m.loop
    m.set_loop(True)
    m.set_enabled(False, True)
    assert m.loop and m.get_enabled() == (False, True)
    def frames_left():
        n = 0
        while m.decode_audio() is not None:
            n += 1
        return n
    m.set_audio_stream(1)                      # a stream the file does not have: only what was buffered before comes out
    assert frames_left() < 40
    m.set_audio_stream(7)                      # out of range: ignored (still stream 1)
    m.rewind()
    assert frames_left() == 0
    m.set_audio_stream(0)
    m.rewind()
    assert frames_left() == 355                # the whole of testdata/test.mpg (mpeg_test.go:164)
    m.close()


def test_seek_keeps_audio_time_in_step(test_mpg, window):
    """mpeg_test.go:402-438: an exact seek (also off a frame boundary) leaves Audio.Time within one packet."""
    times = []
    for ms in (1000, 2000, 3000, 3001, 4000, 5000):
        m = hostlib.HostMpeg(test_mpg, window=window)
        m.count_callbacks()
        assert m.seek(ms / 1000.0, True)
        assert abs(m.audio_time - m.time) <= 0.5, (ms, m.audio_time, m.time)
        times.append(m.audio_time)
        m.close()
    assert abs(times[3] - times[2]) <= 0.5


@pytest.mark.parametrize("exact", [False, True])
def test_seek_fires_the_video_callback_exactly_once(test_mpg, window, exact):
    """mpeg_test.go:442-461."""
    m = hostlib.HostMpeg(test_mpg, window=window)
    m.count_callbacks()
    assert m.seek(3.0, exact)
    assert m.callback_counts()[0] == 1
    m.close()


def test_exact_seek_returns_a_picture_of_the_linear_decode(oracle, test_mpg, window):
    """Beyond the reference's tests: the frame an exact seek returns is bit-identical to a frame of the
    linear decode next to the requested time (decode-forward from the intra frame reproduces the
    references).  Its Time is PTS-based (relative to the lowest PTS, mpeg.go:493), the linear decoder's
    is frame-count based; test.mpg's first packet is two frames above the lowest PTS, hence the slack."""
    lin = hostlib.HostMpeg(test_mpg, window=window)
    lin.set_enabled(True, False)
    frames = []
    while True:
        f = lin.decode_video()
        if f is None:
            break
        frames.append((f.time, [p.copy() for p in hostlib.frame_planes(f)]))
    lin.close()
    assert len(frames) == 278
    m = hostlib.HostMpeg(test_mpg, window=window)
    for t in (2.0, 0.5, 6.25, 3.2, 7.0):
        f = m.seek_frame(t, True)
        assert f is not None and t - 1e-9 <= f.time < t + 1.0 / 30 + 1e-9
        got = hostlib.frame_planes(f)
        same = [ft for ft, planes in frames if all(np.array_equal(a, b) for a, b in zip(planes, got))]
        assert same and abs(same[0] - t) <= 3.0 / 30 + 1e-9, (t, same)
    m.close()


def test_done_audio_format_and_lead_time_accessors(test_mpg, window):
    """mpeg.go:155 (Done: `true` on a channel of capacity 1 when the stream ends without looping, handleEnd :625-632),
    mpeg.go:229-238 (AudioFormat / SetAudioFormat), mpeg.go:301-310 (AudioLeadTime / SetAudioLeadTime)."""
    m = hostlib.HostMpeg(test_mpg, window=window)
    m.count_callbacks()                                      # (Decode only decodes what has a callback, mpeg.go:366-367)
    H = hostlib.host()
    assert H.mpeghost_mpeg_audio_format(m.h) == 0 and H.mpeghost_mpeg_audio_lead_time(m.h) == 0.0    # AudioF32N, no lead
    H.mpeghost_mpeg_set_audio_lead_time(m.h, 0.25)
    H.mpeghost_mpeg_set_audio_format(m.h, 3)
    assert H.mpeghost_mpeg_audio_format(m.h) == 3 and H.mpeghost_mpeg_audio_lead_time(m.h) == 0.25
    H.mpeghost_mpeg_set_audio_format(m.h, 0)
    H.mpeghost_mpeg_set_audio_lead_time(m.h, 0.0)
    assert H.mpeghost_mpeg_take_done(m.h) == 0                # nothing on the channel while the stream runs
    ticks = 0
    while not m.has_ended and ticks < 400:
        m.decode(0.1)
        ticks += 1
    assert m.has_ended
    assert H.mpeghost_mpeg_take_done(m.h) == 1 and H.mpeghost_mpeg_take_done(m.h) == 0   # one value, received once
    m.rewind()
    m.set_loop(True)                                          # looping streams never signal (handleEnd rewinds instead)
    for _ in range(200):
        m.decode(0.1)
    assert not m.has_ended and H.mpeghost_mpeg_take_done(m.h) == 0
    m.close()
