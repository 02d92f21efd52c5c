"""-m gpu: Video.Decode / Audio.Decode one picture / frame ahead on the host (tests/test_host_lookahead.py has the contract), on the
HIP backend: asynchronous read-back into pinned frames (mpeghip_video_read_planes_async), asynchronous synthesis from / into pinned
memory (mpeghip_audio_synth_async) — frames, times, Time(), HasEnded() and what a Rewind leaves behind equal the synchronous form's,
which equals the oracle's (test_gpu_golden.py)."""
import ctypes as C

import numpy as np
import pytest

import hostlib
from test_host_lookahead import SCRIPTS, _view, oracle_script, run_script

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    d = hostlib.host().mpeghost_device_create(0)
    assert d, hostlib.host().mpeghost_last_error()
    yield d
    hostlib.host().mpeghost_device_destroy(d)


@pytest.mark.parametrize("script", SCRIPTS, ids=[str(i) for i in range(len(SCRIPTS))])
def test_video_lookahead_changes_nothing_on_gpu(golden_dir, device, script):
    data = (golden_dir / "test.mpeg1video").read_bytes()
    a, b = hostlib.HostVideo(data, device=device), hostlib.HostVideo(data, device=device)
    b.set_lookahead(False)
    ra, rb = run_script(a, script), run_script(b, script)
    a.close()
    b.close()
    assert ra == rb


@pytest.mark.parametrize("script", SCRIPTS, ids=[str(i) for i in range(len(SCRIPTS))])
def test_video_rewinds_against_the_oracle_on_gpu(oracle, golden_dir, device, script):
    """tests/test_host_lookahead.py::test_video_rewinds_against_the_oracle on the HIP backend (look-ahead and host mirror on): what a
    Rewind leaves in the device's frame store is what the reference's leaves in its three frames (video.go:195-201)"""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    ref, dut = oracle.VideoDecoder(data), hostlib.HostVideo(data, device=device)
    want, got = oracle_script(ref, script, oracle.frame_planes), run_script(dut, script)
    ref.close()
    dut.close()
    assert got == want


def test_video_frames_equal_the_oracles_with_the_lookahead_on(oracle, golden_dir, device):
    """frame by frame, planes + time + Frame.RGBA of the frame in hand (its slot is intact: the next picture is parsed, not submitted)"""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    ref, dut = oracle.VideoDecoder(data), hostlib.HostVideo(data, device=device)
    n = 0
    while True:
        a, b = ref.decode(), dut.decode()
        assert (a is None) == (b is None)
        if a is None:
            break
        for pa, pb in zip(oracle.frame_planes(a), hostlib.frame_planes(b)):
            assert np.array_equal(pa, pb), "frame %d" % n
        assert abs(a.time - b.time) < 1e-12
        if n % 16 == 0:
            want = np.empty((120, 160, 4), np.uint8)
            oracle.lib().orc_ycbcr_to_rgba(C.byref(a), want.ctypes.data)
            assert np.array_equal(want, dut.rgba(160, 120)), "Frame.RGBA of frame %d" % n
        n += 1
    assert n == 260
    ref.close()
    dut.close()


def test_video_frame_is_valid_until_the_next_decode_call_on_gpu(golden_dir, device):
    """without the host mirror: two PINNED frames alternate; the one in the caller's hands does not change while the next call runs"""
    dec = hostlib.HostVideo((golden_dir / "test.mpeg1video").read_bytes(), device=device)
    dec.set_host_mirror(False)
    prev_view, prev_copy, ptrs = None, None, []
    for i in range(40):
        f = dec.decode()
        assert f is not None
        if prev_view is not None:
            assert np.array_equal(prev_view, prev_copy), "frame %d changed during the next decode call" % (i - 1)
        ptrs.append(f.y)
        prev_view, prev_copy = _view(f), _view(f).copy()
    assert len(set(ptrs)) == 2 and ptrs[0::2] == [ptrs[0]] * 20 and ptrs[1::2] == [ptrs[1]] * 20
    dec.close()


def test_video_mirrored_frame_has_the_references_lifetime_on_gpu(golden_dir, device):
    """the default: the returned frame is the slot's copy in the device store's host mirror (mpeghip_video_host_mirror) — one of
    three addresses, as the reference's returned *Frame is one of its three frames; nothing is in flight when Decode returns, so the
    bytes stay as they are until the next decode call begins"""
    import time
    dec = hostlib.HostVideo((golden_dir / "test.mpeg1video").read_bytes(), device=device)
    prev_view, prev_copy, ptrs = None, None, []
    for i in range(60):
        if prev_view is not None:
            if i % 10 == 0:
                time.sleep(0.002)
            assert np.array_equal(prev_view, prev_copy), "frame %d changed before the next decode call" % (i - 1)
        f = dec.decode()
        assert f is not None
        ptrs.append(f.y)
        prev_view, prev_copy = _view(f), _view(f).copy()
    assert len(set(ptrs)) == 3
    dec.close()


@pytest.mark.parametrize("script", SCRIPTS, ids=[str(i) for i in range(len(SCRIPTS))])
def test_video_host_mirror_changes_nothing_on_gpu(golden_dir, device, script):
    data = (golden_dir / "test.mpeg1video").read_bytes()
    a, b = hostlib.HostVideo(data, device=device), hostlib.HostVideo(data, device=device)
    b.set_host_mirror(False)
    ra, rb = run_script(a, script), run_script(b, script)
    a.close()
    b.close()
    assert ra == rb


@pytest.mark.parametrize("fmt", [0, 1, 2, 3], ids=["F32N", "F32NLR", "F32", "S16"])
@pytest.mark.parametrize("script", [[5, "rewind", 7, "rewind", "rewind", 3], [400], [355, 1, "rewind", 2]], ids=["mid", "to_end", "at_end"])
def test_audio_lookahead_changes_nothing_on_gpu(golden_dir, device, fmt, script):
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
                out.append((hash(s[:1152 if fmt == 1 else 2304].tobytes()), dec.time, dec.has_ended))   # (F32NLR: the C API hands out Left)
        return out
    data = (golden_dir / "test.mp2").read_bytes()
    a, b = hostlib.HostAudio(data, device=device, fmt=fmt), hostlib.HostAudio(data, device=device, fmt=fmt)
    b.set_lookahead(False)
    ra, rb = run(a), run(b)
    a.close()
    b.close()
    assert ra == rb and sum(1 for r in ra if r[0] not in (None, "rewind")) >= 3


def test_async_read_back_and_synthesis_through_the_c_abi(oracle, hip_ctx):
    """mpeghip_video_read_planes_async / _read_wait and mpeghip_audio_synth_async / _synth_wait / _undo_last directly: pinned and
    pageable destinations, several read-backs in flight, tickets waited for out of order; a synthesis undone leaves the V ring as
    it was (the next frame comes out as if the undone one had never been launched)."""
    from mpeg_amd import abi, desc, synth
    w, h = 176, 144
    seq = synth.generate_sequence(w, h, 4, seed=9)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)
    n = dut.info.luma_bytes + 2 * dut.info.chroma_bytes
    pins = [hip_ctx.pinned(n) for _ in range(3)]
    for s in seq:
        ref.submit(s.pics, s.mbs, s.coefs)
        dut.submit(s.pics, s.mbs, s.coefs)
        tickets = [dut.read_planes_async(0, slot, pins[slot]) for slot in range(3)]     # three read-backs in flight
        for slot in (2, 0, 1):                                                            # ... waited for out of order
            dut.read_wait(tickets[slot])
            for a, b in zip(ref.read_planes(0, slot), dut.split_planes(pins[slot].u8)):
                assert np.array_equal(a, b)
    # a pageable destination takes the copy path
    page = np.zeros(n, np.uint8)
    t = C.c_uint64()
    assert dut.lib.mpeghip_video_read_planes_async(dut.h, 0, 1, page.ctypes.data_as(C.c_void_p), C.byref(t)) == 0
    dut.read_wait(t.value)
    assert np.array_equal(np.concatenate(ref.read_planes(0, 1)), page)
    with pytest.raises(abi.MpegHipError):
        dut.read_wait(t.value + 1)                                                        # no such read-back
    for p in pins:
        p.free()
    dut.close()
    ref.close()
    # audio
    smp = synth.audio_frames(1, 3)                                                        # [1, 3, 2, 36, 32]
    want = oracle.OracleSynth(1, 0).synth(smp[:, [0, 2]])                                 # frames 0 and 2: frame 1 is undone below
    a = abi.AudioSynth(hip_ctx, 1, desc.AUDIO_FMA_NONE)
    pin_in, pin_out = hip_ctx.pinned(9216), hip_ctx.pinned(9216)
    got = []
    for k in range(3):
        pin_in.view(np.int32)[:] = smp[0, k].reshape(-1)
        t = a.synth_async(pin_in, 1, desc.AUDIO_F32N, pin_out)
        a.synth_wait(t)
        if k == 1:
            a.undo_last()
            with pytest.raises(abi.MpegHipError):
                a.undo_last()                                                             # one level only
        else:
            got.append(pin_out.view(np.float32).copy())
    assert np.array_equal(np.concatenate(got).view(np.uint32), want.reshape(-1).view(np.uint32))
    pin_in.free()
    pin_out.free()
    a.close()
