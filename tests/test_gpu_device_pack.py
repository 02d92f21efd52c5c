"""-m gpu: the DEVICE-PACKED stage (include/mpeghip.h: mpeghip_video_stage_begin_device / _map / _put_mapped, mpeghip_video_sync)
through the C ABI on the MI355X: the host only copies the ABI's arrays, pack_kernel validates and packs them in front of
recon_kernel.  device-packed = host-packed = oracle on seeded sequences (typical, dense, snapshot blocks, fused RGBA, 1080p),
on many streams per commit put from several threads, copied and mapped, on the golden streams and a written 1080p stream
through the parser; every malformed input of the host path's tests refused (deferred, at the next sync) with NOTHING of the
commit reconstructed.  The CPU twin (lane emulator) is tests/test_device_pack_emu.py."""
import numpy as np
import pytest

import hostlib
from mpeg_amd import abi, desc, synth
from parity import assert_planes_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    d = hostlib.host().mpeghost_device_create(0)
    assert d, hostlib.host().mpeghost_last_error()
    yield d
    hostlib.host().mpeghost_device_destroy(d)


@pytest.mark.parametrize("mapped", [False, True], ids=["copied", "mapped"])
@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 7, "typical", 0.1, False),
    (352, 240, 4, "dense", 0.0, False),
    (160, 120, 5, "typical", 0.05, True),
    (50, 35, 4, "typical", 0.1, True),       # a last chunk with dead records, one wave with idle lanes
    (1920, 1080, 4, "typical", 0.01, True),
    (1920, 1080, 3, "dense", 0.0, False),
])
def test_device_packed_equals_host_packed_equals_oracle(oracle, hip_ctx, w, h, n, profile, raw, rgba, mapped):
    seq = synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba, seed=0xD5 + w)
    ref, host_packed, dev_packed = oracle.OracleStore(w, h, threads=4), abi.VideoStore(hip_ctx, w, h), abi.VideoStore(hip_ctx, w, h)
    try:
        for s in seq:
            ref.submit(s.pics, s.mbs, s.coefs)
            mbs, words = desc.to_sparse(s.mbs, s.coefs)
            host_packed.submit_staged_sparse([(s.pics[0], mbs, words)])
            dev_packed.submit_staged_device([(s.pics[0], mbs, words)], mapped=mapped)
            for slot in range(3):
                want = ref.read_planes(0, slot)
                assert_planes_equal(want, dev_packed.read_planes(0, slot), "device-packed, picture type %d slot %d" % (s.picture_type, slot))
                assert_planes_equal(want, host_packed.read_planes(0, slot), "host-packed, picture type %d slot %d" % (s.picture_type, slot))
            if rgba:
                assert np.array_equal(ref.read_rgba(0, s.cur), np.asarray(dev_packed.read_rgba(0, s.cur)).reshape(h, w, 4))
    finally:
        host_packed.close()
        dev_packed.close()
        ref.close()


def test_many_streams_per_commit_from_several_threads(oracle, hip_ctx):
    """Nine streams at different content, one picture each per commit, put by 4 threads — copied, and written through the
    mapping; commits are NOT waited for in between (the verdicts are collected at the end)."""
    w, h, n_streams = 352, 240, 9
    seqs = [synth.generate_sequence(w, h, 6, seed=700 + k, raw_fraction=0.05 if k % 3 == 0 else 0.0,
                                    profile="dense" if k == 4 else "typical", rgba=(k == 7)) for k in range(n_streams)]
    refs = [oracle.OracleStore(w, h) for _ in range(n_streams)]
    copied, mapped = abi.VideoStore(hip_ctx, w, h, n_streams), abi.VideoStore(hip_ctx, w, h, n_streams)
    try:
        for step in range(6):
            parts = []
            for k in range(n_streams):
                s = seqs[k][step]
                refs[k].submit(s.pics, s.mbs, s.coefs)
                pic = s.pics[0].copy()
                pic["stream"] = k
                parts.append((pic,) + desc.to_sparse(s.mbs, s.coefs))
            assert all(rc == 0 for rc in copied.submit_staged_device(parts, threads=4, sync=False))
            assert all(rc == 0 for rc in mapped.submit_staged_device(parts, threads=4, mapped=True, sync=False))
        copied.sync()
        mapped.sync()
        for k in range(n_streams):
            for slot in range(3):
                want = refs[k].read_planes(0, slot)
                assert_planes_equal(want, copied.read_planes(k, slot), "copied: stream %d slot %d" % (k, slot))
                assert_planes_equal(want, mapped.read_planes(k, slot), "mapped: stream %d slot %d" % (k, slot))
    finally:
        copied.close()
        mapped.close()
        for r in refs:
            r.close()


def _cases(good_mbs, words, k):
    """(name, mbs, words, expected code): one kind of damage each — those of test_gpu_sparse / test_gpu_video's refusal tests"""
    out = []

    def case(name, code=abi.ERR_INVALID, **fields):
        m = good_mbs.copy()
        for f, v in fields.items():
            m[f][k] = v
        out.append((name, m, words, code))

    case("mb_x", mb_x=200)
    case("mb_y", mb_y=99)
    case("cbp", cbp=0x40)
    case("qscale 0", qscale=0)
    case("qscale 32", qscale=32)
    case("two references", flags=desc.MB_REF_FWD | desc.MB_REF_BWD)
    case("intra with a reference", flags=desc.MB_INTRA | desc.MB_REF_FWD)
    case("no reference", flags=0)
    case("vector above the plane", code=abi.ERR_RANGE, mv_y=-4000)
    case("vector far below the pad", code=abi.ERR_RANGE, mv_y=4000)
    case("offset beyond the words", coef_off=len(words) + 1)
    m = good_mbs.copy()
    m["mb_x"][k], m["mb_y"][k] = m["mb_x"][k - 1], m["mb_y"][k - 1]
    out.append(("position twice", m, words, abi.ERR_INVALID))
    m = good_mbs.copy()
    m["coef_off"][k + 1] = m["coef_off"][k]
    out.append(("overlapping offsets", m, words, abi.ERR_INVALID))
    m = good_mbs.copy()
    m["coef_off"] = 0
    out.append(("all offsets zero (the round-3 advisor's heap overrun)", m, words, abi.ERR_INVALID))
    at = int(good_mbs["coef_off"][k])
    w2 = words.copy()
    w2[at] = 65
    out.append(("count", good_mbs, w2, abi.ERR_INVALID))
    w2 = words.copy()
    w2[at + 1] |= 0x0100
    out.append(("stray bits", good_mbs, w2, abi.ERR_INVALID))
    out.append(("short", good_mbs, words[:at + 3].copy(), abi.ERR_INVALID))
    return out


def test_malformed_pictures_are_refused_at_the_next_sync_and_the_other_pictures_are_reconstructed(oracle, hip_ctx):
    """Every kind of damage, in a commit of three streams of which ONE is damaged: the commit returns OK, mpeghip_video_verdict /
    mpeghip_video_sync return the error ONCE and mpeghip_video_refused names the picture and its stream; the damaged picture was
    not reconstructed (its stream's slots are untouched), the two healthy streams' pictures WERE (ABI 3: the unit of failure is the
    picture, video.go:374-460), and the handle goes on working — the same pictures, undamaged, then reconstruct bit-exactly.  The
    host-packed stage refuses the same input at its put / commit."""
    w, h, n_streams = 96, 64, 3
    seq = synth.generate_sequence(w, h, 3, seed=8, profile="dense")
    ref = oracle.OracleStore(w, h)
    dut, hostside = abi.VideoStore(hip_ctx, w, h, n_streams), abi.VideoStore(hip_ctx, w, h, n_streams)
    try:
        def parts_of(s, mbs=None, words=None, bad_stream=None):
            gm, gw = desc.to_sparse(s.mbs, s.coefs)
            out = []
            for k in range(n_streams):
                pic = s.pics[0].copy()
                pic["stream"] = k
                out.append((pic, mbs, words) if k == bad_stream else (pic, gm, gw))
            return out

        ref.submit(seq[0].pics, seq[0].mbs, seq[0].coefs)
        dut.submit_staged_device(parts_of(seq[0]))
        hostside.submit_staged_sparse(parts_of(seq[0]))
        s = seq[1]
        good_mbs, good_words = desc.to_sparse(s.mbs, s.coefs)
        k = next(i for i in range(1, len(good_mbs) - 1)
                 if good_mbs[i]["cbp"] and not (good_mbs[i]["flags"] & (desc.MB_COEF_RAW | desc.MB_INTRA)))
        before = [[dut.read_planes(st, slot) for slot in range(3)] for st in range(n_streams)]
        ref.submit(s.pics, s.mbs, s.coefs)                            # (what the healthy streams hold after the commit: the P picture, submitted again and again)
        for i, (name, mbs, words, code) in enumerate(_cases(good_mbs, good_words, k)):
            rcs = dut.submit_staged_device(parts_of(s, mbs, words, bad_stream=1), threads=2, sync=False)   # the commit itself: OK
            assert all(rc == 0 for rc in rcs), name
            with pytest.raises(abi.MpegHipError) as e:
                dut.verdict() if i % 2 else dut.sync()                # (the verdict alone waits for the validation only)
            assert e.value.code == code, (name, str(e.value))
            assert "picture 1 (stream 1)" in str(e.value) and "1 of its 3 pictures refused" in str(e.value), (name, str(e.value))
            assert dut.refused() == (1, [(1, 1)]), name
            dut.verdict()                                             # reported once
            dut.sync()
            for slot in range(3):
                assert_planes_equal(before[1][slot], dut.read_planes(1, slot), "%s: the refused picture's stream, slot %d, was touched" % (name, slot))
                for st in (0, 2):
                    assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(st, slot), "%s: healthy stream %d slot %d" % (name, st, slot))
            with pytest.raises(abi.MpegHipError) as e:                # the host-packed stage: the same input, refused in the call
                hostside.submit_staged_sparse(parts_of(s, mbs, words, bad_stream=1))
            assert e.value.code == code, (name, str(e.value))
        # a verdict nobody asked for surfaces at the call that reuses the staging buffer, or at a read
        name, mbs, words, code = _cases(good_mbs, good_words, k)[0]
        dut.submit_staged_device(parts_of(s, mbs, words, bad_stream=2), sync=False)
        with pytest.raises(abi.MpegHipError):
            dut.read_planes(0, 0)
        dut.read_planes(0, 0)
        # two damaged pictures in one commit: both named, the third picture reconstructed
        dut.submit_staged_device([(p, mbs, words) if st != 0 else (p, good_mbs, good_words)
                                  for st, (p, _, _) in enumerate(parts_of(s))], sync=False)
        with pytest.raises(abi.MpegHipError) as e:
            dut.verdict()
        assert "2 of its 3 pictures refused" in str(e.value) and dut.refused() == (2, [(1, 1), (2, 2)])
        # and the streams continue bit-exactly
        for i, s in enumerate(seq[1:]):
            if i:
                ref.submit(s.pics, s.mbs, s.coefs)
            dut.submit_staged_device(parts_of(s))
        for st in range(n_streams):
            for slot in range(3):
                assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(st, slot), "after the refusals: stream %d slot %d" % (st, slot))
    finally:
        dut.close()
        hostside.close()
        ref.close()


def test_pictures_that_depend_on_each_other_are_refused(oracle, hip_ctx):
    """Two pictures of ONE stream in one commit: writing the same slot is refused by the commit itself (the host sees it);
    a P picture that reads the slot an I picture of the same commit writes is found by the device (only it knows whether a
    macroblock really predicts from that slot)."""
    w, h = 96, 64
    seq = synth.generate_sequence(w, h, 3, seed=12)
    dut = abi.VideoStore(hip_ctx, w, h, 2)
    ref = oracle.OracleStore(w, h)
    try:
        i_pic, p_pic = seq[0], seq[1]
        assert p_pic.picture_type == desc.PIC_P and int(p_pic.pics[0]["fwd"]) == int(i_pic.pics[0]["cur"])
        a = (i_pic.pics[0],) + desc.to_sparse(i_pic.mbs, i_pic.coefs)
        b = (p_pic.pics[0],) + desc.to_sparse(p_pic.mbs, p_pic.coefs)
        with pytest.raises(abi.MpegHipError) as e:
            dut.submit_staged_device([a, a])
        assert e.value.code == abi.ERR_INVALID and "depend" in str(e.value)
        dut.submit_staged_device([a, b], sync=False)
        with pytest.raises(abi.MpegHipError) as e:
            dut.sync()
        assert e.value.code == abi.ERR_INVALID and "depend" in str(e.value) and "picture 1" in str(e.value)
        assert dut.refused() == (1, [(1, 0)])
        # the P picture — the one that reads what the other writes — was refused, the I picture reconstructed
        ref.submit(i_pic.pics, i_pic.mbs, i_pic.coefs)
        for slot in range(3):
            assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(0, slot), "after the refused P picture: slot %d" % slot)
        # the same two pictures in two commits
        ref.submit(p_pic.pics, p_pic.mbs, p_pic.coefs)
        dut.submit_staged_device([a])
        dut.submit_staged_device([b])
        for slot in range(3):
            assert_planes_equal(ref.read_planes(0, slot), dut.read_planes(0, slot), "slot %d" % slot)
    finally:
        dut.close()
        ref.close()


def test_a_device_packed_stage_takes_sparse_pictures_only(hip_ctx):
    w, h = 96, 64
    s = synth.generate_sequence(w, h, 1, seed=1)[0]
    dut = abi.VideoStore(hip_ctx, w, h)
    lib = dut.lib
    import ctypes as C
    try:
        n_mbs, n_words = np.array([len(s.mbs)], np.uint32), np.array([s.coefs.nbytes // 4], np.uint64)
        st = C.c_void_p()
        assert lib.mpeghip_video_stage_begin_device(dut.h, 1, abi._ptr(n_mbs), abi._ptr(n_words), C.byref(st)) == 0
        pic, mbs, coefs = dut._args(s.pics, s.mbs, s.coefs)
        assert lib.mpeghip_video_stage_put(st, 0, abi._ptr(pic), abi._ptr(mbs), abi._ptr(coefs)) == abi.ERR_INVALID
        assert lib.mpeghip_video_stage_commit(st) == abi.ERR_INVALID          # the failed put's error, nothing launched
        pm, pw = C.c_void_p(), C.c_void_p()
        st2 = C.c_void_p()
        assert lib.mpeghip_video_stage_begin_sparse(dut.h, 1, abi._ptr(n_mbs), abi._ptr(n_words), C.byref(st2)) == 0
        assert lib.mpeghip_video_stage_map(st2, 0, C.byref(pm), C.byref(pw)) == abi.ERR_INVALID   # not a device-packed stage
        assert lib.mpeghip_video_stage_commit(st2) == abi.ERR_INVALID         # picture 0 was never put
    finally:
        dut.close()


def test_golden_and_written_streams_through_the_parser_and_the_device_packer(oracle, golden_dir, device):
    """VideoBatch with a thread pool on the HIP store: its staged submits are device-packed by default — the damaged golden
    stream (snapshot blocks, invalid intra blocks, re-submits), four streams at different GOP phases; and with the packing left
    to the host: the same hashes."""
    from test_host_batch import VIDEO_HASH, run_batch
    es = (golden_dir / "test.mpeg1video").read_bytes()
    for device_pack in (True, False):
        h, n, c = run_batch(oracle, [es] * 4, [0, 0, 1, 5], device=device, threads=3, device_pack=device_pack)
        assert h == [VIDEO_HASH] * 4 and n == [260] * 4, device_pack


@pytest.mark.parametrize("fetch_around", [True, False], ids=["fetch", "frames_stay_on_the_device"])
def test_a_refused_picture_is_reported_by_the_next_call_on_gpu(oracle, golden_dir, device, fetch_around):
    """tests/test_host_batch.py's contract on the HIP store: pack_gate_kernel refuses the damaged picture alone, the next DecodeAll
    reports it (mpeghip_video_verdict: before its own round reaches the device), the other streams' frames are the golden ones
    to the end."""
    from test_host_batch import run_refusal
    run_refusal(oracle, (golden_dir / "test.mpeg1video").read_bytes(), device, 3, fetch_around)


def test_the_damaged_golden_stream_on_recon_kernel_too(oracle, golden_dir, device):
    """A few streams of 160x120 are launches of a few dozen chunks, which the library gives to recon_wide_kernel; 112 lockstep
    copies of the damaged golden stream are 2 240 chunks per tick — more than four waves per chunk fit — so that ITS snapshot
    blocks, invalid intra blocks, chunks that are not runs and re-submits run through recon_kernel's one-wave-per-chunk instances
    as well (both hand-overs).  Every stream: the reference's hash."""
    from test_host_batch import VIDEO_HASH, run_batch
    es = (golden_dir / "test.mpeg1video").read_bytes()
    for device_pack in (True, False):
        h, n, c = run_batch(oracle, [es] * 112, [0] * 112, device=device, threads=4, device_pack=device_pack)
        assert h == [VIDEO_HASH] * 112 and n == [260] * 112, device_pack


def test_config5_shard_device_packed(oracle, hip_ctx):
    """One picture for each of 256 1080p streams per commit, staged from 8 threads and packed on the device: a GOP's worth,
    every stream x slot against the oracle by device-side hash."""
    w, h, n_streams = 1920, 1080, 256
    seq = synth.generate_sequence(w, h, 5, seed=0xC5)
    ref, dut = oracle.OracleStore(w, h, threads=8), abi.VideoStore(hip_ctx, w, h, n_streams)
    try:
        for s in seq:
            ref.submit(s.pics, s.mbs, s.coefs)
            mbs, words = desc.to_sparse(s.mbs, s.coefs)
            parts = []
            for k in range(n_streams):
                pic = s.pics[0].copy()
                pic["stream"] = k
                parts.append((pic, mbs, words))
            dut.submit_staged_device(parts, threads=8, sync=False)
        dut.sync()
        for slot in range(3):
            want = oracle.FNV_OFFSET
            for p in ref.read_planes(0, slot):
                want = oracle.fnv1a64(p, want)
            assert (dut.hash_slots(slot) == np.uint64(want)).all(), "slot %d" % slot
    finally:
        dut.close()
        ref.close()
