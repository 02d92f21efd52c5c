"""-m gpu: the host mirror (mpeghip_video_host_mirror / _mirror_async, include/mpeghip.h): the reconstruction launch of a small
submit writes every macroblock once more, linearly, into the frame's copy in pinned host memory (recon_wide_kernel<false, true>,
rc_mirror_mb) — the copy must be the slot's linear planes (= the oracle's frame) after every picture, through runs, chunks that are
not runs, intra macroblocks with blocks that keep the old pixels, pictures that cover part of the frame, several streams in one
launch; and whatever else writes a slot (a launch of the other kernel, a colour-converting picture, write_planes) must be noticed
and repaired by the next request."""
import numpy as np
import pytest

from parity import assert_planes_equal

pytestmark = pytest.mark.gpu


def damage(seq, rng):
    """intra macroblocks lose a coded block (it keeps the old pixels: video.go:711-714 as the emitter hands it over) and every
    third picture loses a tenth of its macroblocks (positions nothing writes; chunks whose macroblocks are not consecutive)"""
    from mpeg_amd import desc
    for i, s in enumerate(seq):
        intra = np.nonzero((s.mbs["flags"] & desc.MB_INTRA) != 0)[0]
        for k in intra[rng.random(len(intra)) < 0.3]:
            cbp = int(s.mbs["cbp"][k])
            s.mbs["cbp"][k] = cbp & (cbp - 1)          # the LAST coded block goes: the others keep their units
        if i % 3 == 2:
            keep = rng.random(len(s.mbs)) >= 0.1
            s.mbs = s.mbs[keep].copy()
            s.pics["mb_count"] = len(s.mbs)
    return seq


def check_mirrors(ref, dut, stream, what):
    views = [dut.mirror_async(stream, slot) for slot in range(3)]
    for slot in (1, 2, 0):                               # waited for out of order
        view, ticket = views[slot]
        dut.read_wait(ticket)
        assert_planes_equal(ref.read_planes(stream, slot), dut.split_planes(view), "%s slot %d (host mirror)" % (what, slot))
        assert_planes_equal(ref.read_planes(stream, slot), dut.read_planes(stream, slot), "%s slot %d (frame store)" % (what, slot))


@pytest.mark.parametrize("w,h", [(352, 240), (160, 120), (48, 32)], ids=["sif", "160x120", "3x2"])
@pytest.mark.parametrize("sparse", [False, True], ids=["units", "sparse"])
def test_mirror_is_the_frame_after_every_picture(oracle, hip_ctx, w, h, sparse):
    from mpeg_amd import abi, desc, synth
    rng = np.random.default_rng(w * 7 + sparse)
    seq = damage(synth.generate_sequence(w, h, 13, seed=w + 3, raw_fraction=0.04), rng)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)
    dut.host_mirror()
    dut.host_mirror()                                    # (asking twice changes nothing)
    for i, s in enumerate(seq):
        ref.submit(s.pics, s.mbs, s.coefs)
        if sparse:
            mbs, words = desc.to_sparse(s.mbs, s.coefs)
            dut.submit_sparse(s.pics[0], mbs, words)
        else:
            dut.submit(s.pics, s.mbs, s.coefs)
        check_mirrors(ref, dut, 0, "picture %d (type %d)" % (i, s.picture_type))
    # every launch took the mirroring instance: the three slots were untiled once each (their first request), never again
    assert dut.mirror_counters() == (3 * len(seq), 3)
    dut.close()
    ref.close()


def test_other_writers_are_noticed_and_repaired(oracle, hip_ctx):
    """pinned tile policies name recon_kernel's instances (no mirror code); MPEGHIP_PIC_RGBA pictures take the colour-converting
    instance; write_planes writes a slot directly — after each of them the next request must still hand out the slot's planes"""
    from mpeg_amd import abi, desc, synth
    w, h = 176, 144
    seq = synth.generate_sequence(w, h, 12, seed=21)
    ref, dut = oracle.OracleStore(w, h), abi.VideoStore(hip_ctx, w, h)
    dut.host_mirror()
    for i, s in enumerate(seq):
        if i % 4 == 1:
            dut.set_tile_policy(1 + (i // 4) % 2)         # this picture through recon_kernel
        if i % 4 == 2:
            s.pics["flags"] |= desc.PIC_RGBA              # this one through the colour-converting wide kernel
        ref.submit(s.pics, s.mbs, s.coefs)
        dut.submit(s.pics, s.mbs, s.coefs)
        dut.set_tile_policy(0)
        repairs = dut.mirror_counters()[1]
        check_mirrors(ref, dut, 0, "picture %d" % i)
        if i >= 4:   # (by then every slot has been asked for): the slot one of the other writers wrote is untiled, and only that one
            assert dut.mirror_counters()[1] - repairs == (1 if i % 4 in (1, 2) else 0), i
        if i % 4 == 3:                                    # ... and a slot written from the host
            y, cb, cr = (np.full_like(p, 17 * (k + 1)) for k, p in enumerate(ref.read_planes(0, s.cur)))
            ref.write_planes(0, s.cur, y, cb, cr)
            dut.write_planes(0, s.cur, y, cb, cr)
            check_mirrors(ref, dut, 0, "after write_planes, picture %d" % i)
    dut.host_mirror(False)                                # switched off: the entry refuses, everything else works on
    with pytest.raises(abi.MpegHipError):
        dut.mirror_async(0, 0)
    assert_planes_equal(ref.read_planes(0, 0), dut.read_planes(0, 0))
    dut.close()
    ref.close()


def test_several_streams_in_one_launch_and_the_size_limit(oracle, hip_ctx):
    from mpeg_amd import abi, desc, synth
    w, h, n = 96, 64, 5
    seqs = [synth.generate_sequence(w, h, 6, seed=40 + st) for st in range(n)]
    ref, dut = oracle.OracleStore(w, h, n), abi.VideoStore(hip_ctx, w, h, n)
    dut.host_mirror()
    for i in range(6):
        pics, mbs, coefs, mb0, c0 = [], [], [], 0, 0
        for st in range(n):                               # one submit: every stream's picture, offsets rebased
            s = seqs[st][i]
            p, m = s.pics.copy(), s.mbs.copy()
            p["stream"], p["mb_first"] = st, mb0
            m["pic"] = st
            m["coef_off"] += c0
            pics.append(p)
            mbs.append(m)
            coefs.append(s.coefs)
            mb0 += len(m)
            c0 += len(s.coefs) // desc.COEF_UNIT
        pics, mbs, coefs = np.concatenate(pics), np.concatenate(mbs), np.concatenate(coefs)
        ref.submit(pics, mbs, coefs)
        dut.submit(pics, mbs, coefs)
        for st in range(n):
            check_mirrors(ref, dut, st, "picture %d stream %d" % (i, st))
    dut.close()
    ref.close()
    big = abi.VideoStore(hip_ctx, 1920, 1080, 128)        # 128 x 3 x 3.1 MB > 1 GiB: not what pinned host memory is for
    with pytest.raises(abi.MpegHipError):
        big.host_mirror()
    big.close()
