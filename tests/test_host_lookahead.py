"""Video.Decode / Audio.Decode work ONE PICTURE / FRAME AHEAD on the host (mpeg.hpp, round 6): a call hands the picture parsed
during the previous call to the device, queues the read-back of the frame it returns, parses the next picture while the device
works and only then waits.  What must not change: the frames, their times, Time() / HasEnded() as the reference has them
(video.go:183-207), the lifetime of the returned pointers (mpeg.go:413-415: valid until the next decode call), and everything a
Rewind leaves behind — a picture parsed ahead has not reached the device, so dropping it gives the reference's state back.
CPU: the parser drives the test-only lane emulator; the same sequences run on the HIP backend under -m gpu (test_gpu_lookahead.py)."""
import ctypes as C

import numpy as np
import pytest

import hostlib


def _planes(f):
    return np.concatenate(hostlib.frame_planes(f))


def _view(f):
    """the decoder's own bytes, no copy"""
    n = f.luma_bytes + 2 * f.chroma_bytes
    return np.ctypeslib.as_array(C.cast(f.y, C.POINTER(C.c_uint8)), shape=(n,))


def run_script(dec, script):
    """script: ints = decode that many frames, "rewind"; -> list of (planes hash, time, Time() after, HasEnded() after)"""
    out = []
    for step in script:
        if step == "rewind":
            dec.rewind()
            out.append(("rewind", dec.time, dec.has_ended))
            continue
        for _ in range(step):
            f = dec.decode()
            if f is None:
                out.append((None, dec.time, dec.has_ended))
                break
            out.append((hash(_planes(f).tobytes()), f.time, dec.time, dec.has_ended))
    return out


SCRIPTS = [
    [5, "rewind", 7, "rewind", "rewind", 3],          # rewinds in the middle of a GOP, twice in a row
    [1, "rewind", 1, "rewind", 2],                    # right after the first frame (the first reference picture yields none)
    [400],                                            # through the end of the stream: the flushed last reference picture, then None
    [259, "rewind", 4],                               # one frame before the end, then again from the top
    [260, 1, "rewind", 2],                            # the end itself: None, then rewind
]


@pytest.mark.parametrize("script", SCRIPTS, ids=[str(i) for i in range(len(SCRIPTS))])
def test_video_lookahead_changes_nothing(golden_dir, script):
    """The same calls with the look-ahead on (the default) and off (parse, submit, read back — the form the golden hashes pinned):
    identical frames, frame times, Time(), HasEnded() after every call, across Rewinds."""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    a, b = hostlib.HostVideo(data, emu_flavour=0), hostlib.HostVideo(data, emu_flavour=0)
    b.set_lookahead(False)
    ra, rb = run_script(a, script), run_script(b, script)
    a.close()
    b.close()
    assert ra == rb
    assert sum(1 for r in ra if r[0] not in (None, "rewind")) >= 3


def test_video_frame_is_valid_until_the_next_decode_call(golden_dir):
    """mpeg.go:413-415.  The frame returned by call n keeps its bytes while the caller holds it — although the decoder has, by
    then, parsed the next picture — and call n + 1 returns ANOTHER buffer (two alternate), so even a caller that looks at frame n
    during call n + 1 (a callback on another thread) does not see it change under its eyes before that call returns."""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    dec = hostlib.HostVideo(data, emu_flavour=0)
    prev_view, prev_copy, ptrs = None, None, []
    for i in range(30):
        f = dec.decode()
        assert f is not None
        if prev_view is not None:
            assert np.array_equal(prev_view, prev_copy), "frame %d changed during the next decode call" % (i - 1)
            assert f.y != ptrs[-1]
        ptrs.append(f.y)
        prev_view, prev_copy = _view(f), _view(f).copy()
    assert len(set(ptrs)) == 2 and ptrs[0::2] == [ptrs[0]] * 15 and ptrs[1::2] == [ptrs[1]] * 15
    dec.close()


@pytest.mark.parametrize("script", SCRIPTS[:3], ids=["0", "1", "2"])
@pytest.mark.parametrize("lookahead", [True, False], ids=["ahead", "plain"])
def test_video_host_mirror_changes_nothing(golden_dir, script, lookahead):
    """Decode's frames out of the backend's HOST MIRROR (VideoBackend::mirrorAsync — on the device: mpeghip_video_host_mirror, the
    reconstruction launch writes every frame once more, linearly, into pinned host memory; here: the lane emulator's
    rc_mirror_mb through emu_wide_chunk) against the read-back: identical frames, times, Time(), HasEnded(), across Rewinds —
    through the damaged golden stream's partial pictures and the blocks that keep their old pixels."""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    a, b = hostlib.HostVideo(data, emu_flavour=0), hostlib.HostVideo(data, emu_flavour=0)
    a.set_host_mirror(True)     # (the emulator backend's mirror is off until asked for; the HIP backend's is on by default)
    b.set_host_mirror(False)
    for d in (a, b):
        d.set_lookahead(lookahead)
    ra, rb = run_script(a, script), run_script(b, script)
    a.close()
    b.close()
    assert ra == rb
    assert sum(1 for r in ra if r[0] not in (None, "rewind")) >= 3


def test_video_mirrored_frame_has_the_references_lifetime(golden_dir):
    """With the host mirror the returned frame IS the slot's copy — as in the reference, where the returned *Frame is one of the
    decoder's three (video.go:247-256): it keeps its bytes until the next decode call begins (the picture parsed ahead has not been
    handed over), consecutive B frames come back at the same address, and there are three addresses in all."""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    dec = hostlib.HostVideo(data, emu_flavour=0)
    dec.set_host_mirror(True)
    ptrs, prev_view, prev_copy = [], None, None
    for i in range(40):
        if prev_view is not None:
            assert np.array_equal(prev_view, prev_copy), "frame %d changed before the next decode call" % (i - 1)
        f = dec.decode()
        assert f is not None
        ptrs.append(f.y)
        prev_view, prev_copy = _view(f), _view(f).copy()
    assert len(set(ptrs)) == 3
    dec.close()


def oracle_script(dec, script, planes):
    """run_script's sequence on the ORACLE's decoder (oracle/mpeg_oracle.c: orc_video_rewind restates video.go:195-201)"""
    out = []
    for step in script:
        if step == "rewind":
            dec.rewind()
            out.append(("rewind", dec.time, dec.has_ended))
            continue
        for _ in range(step):
            f = dec.decode()
            if f is None:
                out.append((None, dec.time, dec.has_ended))
                break
            out.append((hash(np.concatenate(planes(f)).tobytes()), f.time, dec.time, dec.has_ended))
    return out


@pytest.mark.parametrize("script", SCRIPTS, ids=[str(i) for i in range(len(SCRIPTS))])
@pytest.mark.parametrize("mirror", [False, True], ids=["read_back", "mirror"])
def test_video_rewinds_against_the_oracle(oracle, golden_dir, script, mirror):
    """Rewind keeps the three frames' bytes and the rotation (video.go:195-201 resets the buffer, the time, hasReferenceFrame and
    the start code only): what the damaged golden stream's first pictures predict from after a rewind in mid-stream is what was
    decoded before it — the whole stream then hashes to something else than a fresh decoder's.  The product one picture ahead
    (the picture parsed ahead never reached the device), with and without the host mirror, against the restated reference: frames,
    frame times, Time() and HasEnded() after every call."""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    ref, dut = oracle.VideoDecoder(data), hostlib.HostVideo(data, emu_flavour=0)
    dut.set_host_mirror(mirror)
    want, got = oracle_script(ref, script, oracle.frame_planes), run_script(dut, script)
    ref.close()
    dut.close()
    assert got == want


@pytest.mark.parametrize("script", [[5, "rewind", 7, "rewind", "rewind", 3], [1, "rewind", 2], [400], [355, 1, "rewind", 2]],
                         ids=["mid", "first", "to_end", "at_end"])
def test_audio_rewinds_against_the_oracle(oracle, golden_dir, script):
    """audio.go:149-154: Rewind does not clear the V ring — the first frames after it are synthesised on top of what the frames
    RETURNED before it left behind (a frame parsed ahead was never synthesised)."""
    def run(dec, bits):
        out = []
        for step in script:
            if step == "rewind":
                dec.rewind()
                out.append(("rewind", dec.time, dec.has_ended))
                continue
            for _ in range(step):
                s = dec.decode()
                if s is None:
                    out.append((None, dec.time, dec.has_ended))
                    break
                out.append((hash(bits(s)), dec.time, dec.has_ended))
        return out
    ref, dut = oracle.AudioDecoder((golden_dir / "test.mp2").read_bytes(), 0), _audio(golden_dir)
    want = run(ref, lambda s: np.asarray(s, np.float32).tobytes())
    got = run(dut, lambda s: np.asarray(s, np.float32)[:2304].tobytes())
    ref.close()
    dut.close()
    assert got == want


def test_video_rgba_of_the_returned_frame_with_a_picture_parsed_ahead(oracle, golden_dir):
    """Frame.RGBA() converts the returned frame's slot on the device: the picture parsed ahead has not been handed over, so the
    slot still holds that frame (B pictures follow one another in ONE slot)."""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    ref, dut = oracle.VideoDecoder(data), hostlib.HostVideo(data, emu_flavour=0)
    for i in range(12):
        a, b = ref.decode(), dut.decode()
        want = np.empty((120, 160, 4), np.uint8)
        oracle.lib().orc_ycbcr_to_rgba(C.byref(a), want.ctypes.data)
        assert np.array_equal(want, dut.rgba(160, 120)), "Frame.RGBA of frame %d" % i
    ref.close()
    dut.close()


def _audio(golden_dir, fmt=0):
    import emu
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    return hostlib.HostAudio((golden_dir / "test.mp2").read_bytes(), fma=0, fmt=fmt, window=win)


@pytest.mark.parametrize("script", [[5, "rewind", 7, "rewind", "rewind", 3], [1, "rewind", 2], [400], [355, 1, "rewind", 2]],
                         ids=["mid", "first", "to_end", "at_end"])
def test_audio_lookahead_changes_nothing(golden_dir, script):
    """As for video; and the V ring, which Rewind keeps (audio.go:149-154), is what the frames RETURNED left behind: a frame parsed
    ahead was never synthesised, so the first frames after a Rewind come out bit-identical with and without the look-ahead."""
    def run(dec):
        out = []
        for step in script:
            if step == "rewind":
                dec.rewind()
                out.append(("rewind", dec.time, dec.has_ended))
                continue
            for _ in range(step):
                s = dec.decode()
                if s is None:
                    out.append((None, dec.time, dec.has_ended))
                    break
                out.append((hash(s.tobytes()), dec.time, dec.has_ended))
        return out
    a, b = _audio(golden_dir), _audio(golden_dir)
    b.set_lookahead(False)
    ra, rb = run(a), run(b)
    a.close()
    b.close()
    assert ra == rb and sum(1 for r in ra if r[0] not in (None, "rewind")) >= 3


def test_audio_samples_are_valid_until_the_next_decode_call(golden_dir):
    """mpeg.go:435-437: two Samples alternate."""
    dec = _audio(golden_dir)
    ptrs, prev, prev_copy = [], None, None
    for i in range(20):
        r = dec.decode_view()
        assert r is not None
        view = np.ctypeslib.as_array(C.cast(r[0], C.POINTER(C.c_float)), shape=(2304,))
        if prev is not None:
            assert np.array_equal(prev.view(np.uint32), prev_copy.view(np.uint32)), "samples %d changed during the next decode call" % (i - 1)
        ptrs.append(r[0])
        prev, prev_copy = view, view.copy()
    assert len(set(ptrs)) == 2 and ptrs[0::2] == [ptrs[0]] * 10
    dec.close()
