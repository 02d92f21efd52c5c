"""-m gpu: the sparse hand-over (include/mpeghip.h: MPEGHIP_PIC_SPARSE, mpeghip_video_stage_begin_sparse / _put_sparse /
mpeghip_video_submit_sparse) through the C ABI on the MI355X: sparse = units = oracle, on seeded descriptor sequences (typical,
dense, snapshot blocks, fused RGBA), on pictures of both forms in ONE staged submit put from several threads, on the
golden streams and a written 1080p stream through the parser in both of its forms; the refusals."""
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


@pytest.mark.parametrize("tile", [0, 1, 2], ids=["auto", "int16", "int32"])
@pytest.mark.parametrize("w,h,n,profile,raw,rgba", [
    (352, 240, 7, "typical", 0.1, False),
    (352, 240, 4, "dense", 0.0, False),
    (160, 120, 5, "typical", 0.05, True),
    (1920, 1080, 4, "typical", 0.0, True),
])
def test_sparse_submit_equals_units_equals_oracle(oracle, hip_ctx, w, h, n, profile, raw, rgba, tile):
    seq = synth.generate_sequence(w, h, n, profile=profile, raw_fraction=raw, rgba=rgba, seed=0x5AA5 + w)
    ref, units, sparse = oracle.OracleStore(w, h, threads=4), abi.VideoStore(hip_ctx, w, h), abi.VideoStore(hip_ctx, w, h)
    units.set_tile_policy(tile)
    sparse.set_tile_policy(tile)
    try:
        for s in seq:
            ref.submit(s.pics, s.mbs, s.coefs)
            units.submit(s.pics, s.mbs, s.coefs)
            mbs, words = desc.to_sparse(s.mbs, s.coefs)
            sparse.submit_sparse(s.pics[0], mbs, words)
            for slot in range(3):
                want = ref.read_planes(0, slot)
                assert_planes_equal(want, sparse.read_planes(0, slot), "sparse, picture type %d slot %d" % (s.picture_type, slot))
                assert_planes_equal(want, units.read_planes(0, slot), "units, picture type %d slot %d" % (s.picture_type, slot))
            if rgba:
                assert np.array_equal(ref.read_rgba(0, s.cur), np.asarray(sparse.read_rgba(0, s.cur)).reshape(h, w, 4))
    finally:
        units.close()
        sparse.close()
        ref.close()


def test_pictures_of_both_forms_in_one_staged_submit_from_several_threads(oracle, hip_ctx):
    """Six streams, one picture each per submit, odd streams sparse, even streams in units, put by 4 threads; and the
    same through mpeghip_video_stage_begin_sparse / _put_sparse."""
    w, h, n_streams = 352, 240, 6
    seqs = [synth.generate_sequence(w, h, 5, seed=900 + k, raw_fraction=0.05 if k % 3 == 0 else 0.0,
                                    profile="dense" if k == 4 else "typical") for k in range(n_streams)]
    refs = [oracle.OracleStore(w, h) for _ in range(n_streams)]
    mixed, all_sparse = abi.VideoStore(hip_ctx, w, h, n_streams), abi.VideoStore(hip_ctx, w, h, n_streams)
    try:
        for step in range(5):
            parts, sparse_parts = [], []
            for k in range(n_streams):
                s = seqs[k][step]
                refs[k].submit(s.pics, s.mbs, s.coefs)
                pic = s.pics[0].copy()
                pic["stream"] = k
                mbs, words = desc.to_sparse(s.mbs, s.coefs)
                sparse_parts.append((pic, mbs, words))
                if k % 2:
                    p2 = pic.copy()
                    p2["flags"] |= desc.PIC_SPARSE
                    parts.append((p2, mbs, words.view(np.uint8)))
                else:
                    parts.append((pic, s.mbs, s.coefs))
            assert all(rc == 0 for rc in mixed.submit_staged(parts, threads=4))
            assert all(rc == 0 for rc in all_sparse.submit_staged_sparse(sparse_parts, threads=4))
            for k in range(n_streams):
                for slot in range(3):
                    want = refs[k].read_planes(0, slot)
                    assert_planes_equal(want, mixed.read_planes(k, slot), "mixed forms: step %d stream %d slot %d" % (step, k, slot))
                    assert_planes_equal(want, all_sparse.read_planes(k, slot), "sparse: step %d stream %d slot %d" % (step, k, slot))
    finally:
        mixed.close()
        all_sparse.close()
        for r in refs:
            r.close()


def test_a_replicated_batch_of_sparse_pictures(oracle, hip_ctx):
    """Benchmark batches (upload once, replicate on the device, replay) take sparse pictures like any other entry point."""
    w, h, n_streams = 352, 240, 33
    seq = synth.generate_sequence(w, h, 4, seed=31)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h, n_streams)
    try:
        for s in seq:
            ref.submit(s.pics, s.mbs, s.coefs)
            mbs, words = desc.to_sparse(s.mbs, s.coefs)
            pics = s.pics.copy()
            pics["flags"] |= desc.PIC_SPARSE
            b = dut.upload(pics, mbs, words.view(np.uint8), replicate=n_streams)
            b.run()
            b.free()
        for slot in range(3):
            want = oracle.FNV_OFFSET
            for p in ref.read_planes(0, slot):
                want = oracle.fnv1a64(p, want)
            assert (dut.hash_slots(slot) == np.uint64(want)).all()
    finally:
        dut.close()
        ref.close()


@pytest.mark.parametrize("damage", ["count", "stray", "short", "dc"])
def test_malformed_sparse_pictures_are_refused_and_nothing_is_launched(hip_ctx, damage):
    w, h = 96, 64
    s = synth.generate_sequence(w, h, 1, seed=8, profile="dense" if damage != "dc" else "typical")[0]
    mbs, words = desc.to_sparse(s.mbs, s.coefs)
    words = [int(x) for x in words]
    k = next(i for i, m in enumerate(mbs) if m["cbp"] and not (m["flags"] & desc.MB_COEF_RAW) and (damage != "dc" or (m["flags"] & desc.MB_INTRA)))
    at = int(mbs[k]["coef_off"])
    if damage == "count":
        words[at] = 65
    elif damage == "stray":
        words[at + 1] |= 0x0100
    elif damage == "short":
        del words[at + 3:]
    else:
        words[at + 1] |= 5 << 2
    dut = abi.VideoStore(hip_ctx, w, h)
    try:
        before = [p.copy() for p in dut.read_planes(0, int(s.cur))]
        with pytest.raises(abi.MpegHipError) as e:
            dut.submit_sparse(s.pics[0], mbs, np.array(words, np.uint32))
        assert e.value.code == -1 and "sparse" in str(e.value)   # MPEGHIP_ERR_INVALID
        for a, b in zip(before, dut.read_planes(0, int(s.cur))):
            assert np.array_equal(a, b)
    finally:
        dut.close()


@pytest.mark.parametrize("sparse", [0, 1], ids=["units", "sparse"])
def test_golden_and_written_streams_through_the_parser_in_both_forms(oracle, golden_dir, device, sparse):
    """The parser's two hand-over forms through the HIP backend: the damaged golden stream (single decoder and four
    streams through VideoBatch with three threads: staged puts), and a written 1080p stream."""
    import mpeg1_writer
    from test_host_batch import VIDEO_HASH, run_batch
    from test_host_parser import video_hash
    from test_written_streams import decode_all, expected_frames
    hostlib.host().mpeghost_set_default_sparse(sparse)
    try:
        es = (golden_dir / "test.mpeg1video").read_bytes()
        dec = hostlib.HostVideo(es, device=device)
        assert video_hash(oracle, dec) == (VIDEO_HASH, 260)
        dec.close()
        h, n, c = run_batch(oracle, [es] * 4, [0, 0, 1, 5], device=device, threads=3)
        assert h == [VIDEO_HASH] * 4 and n == [260] * 4
        w, hh = 1920, 1080
        seq = synth.generate_sequence(w, hh, 4, seed=0x1081)
        want = expected_frames(oracle, w, hh, seq)
        dut = hostlib.HostVideo(mpeg1_writer.write_sequence(w, hh, seq), device=device)
        got = decode_all(dut, hostlib.frame_planes)
        dut.close()
        assert len(got) == len(want)
        for i, (a, b) in enumerate(zip(want, got)):
            for pa, pb in zip(a, b):
                assert np.array_equal(pa, pb), "frame %d" % i
    finally:
        hostlib.host().mpeghost_set_default_sparse(1)
