"""Randomised differential soak of the reconstruction path on the GPU: seeded (geometry, content, picture types, snapshot
share, fused RGBA, kernel policy, the host mirror, hand-over form: dense units / sparse words packed on the host / sparse words packed on the device) cases beyond the fixed parametrisations of tests/, every picture of every case compared with
the oracle on all three slots (and the RGBA image when fused) through the C ABI.  Test infrastructure: the oracle is the checker.
A second phase does the same for the MP2 synthesis (stream counts, frames per call, calls in a row on one state, the four output
formats, both window arithmetics, streams masked out of a call): bit equality with the oracle's synthesis and of the V ring state.
usage: python tools/gpu_soak.py [seconds] [master seed] [audio seconds]  -> a summary line per 25 cases, exit status 1 on any mismatch"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Via:
    """the device store behind one of the three hand-over forms"""
    def __init__(self, store, form, mirror=False):
        self.store, self.form, self.mirror = store, form, mirror
        self.read_rgba = store.read_rgba
        if mirror:
            store.host_mirror()

    def read_planes(self, stream, slot):
        """the slot's planes; with the host mirror on (mpeghip_video_host_mirror) its copy in pinned host memory — written by the
        launch itself where the library's four-waves-per-chunk kernel ran without a colour conversion, repaired by an untiling
        launch elsewhere — must say the same"""
        planes = self.store.read_planes(stream, slot)
        if self.mirror:
            view, ticket = self.store.mirror_async(stream, slot)
            self.store.read_wait(ticket)
            assert np.array_equal(np.concatenate(planes), view), "the host mirror of slot %d differs from the frame store" % slot
        return planes

    def submit(self, pics, mbs, coefs):
        from mpeg_amd import desc
        if self.form == 0:
            return self.store.submit(pics, mbs, coefs)
        m, w = desc.to_sparse(mbs, coefs)
        if self.form == 1:
            return self.store.submit_sparse(pics[0], m, w)
        return self.store.submit_staged_device([(pics[0], m, w)], mapped=bool(len(m) & 1))


def video_case(ctx, rng, allow_big=True):
    """One seeded case, drawn from `rng`, through the device and the oracle; AssertionError (with the case's parameters) on a
    mismatch.  -> (pictures, macroblocks, policy, form).  tests/test_gpu_soak.py runs a fixed number of them under -m gpu."""
    from mpeg_amd import abi, synth
    from oracle import pyoracle
    from parity import run_and_compare
    big = allow_big and rng.random() < 0.08
    w = int(rng.integers(16, 1921 if big else 420))
    h = int(rng.integers(16, 1089 if big else 300))
    n = int(rng.integers(2, 4 if big else 9))
    profile = "dense" if rng.random() < 0.3 else "typical"
    raw = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
    rgba = bool(rng.random() < 0.5)
    policy = int(rng.integers(0, 3))
    form = int(rng.integers(0, 3))
    seed = int(rng.integers(1, 1 << 30))
    types = None
    if rng.random() < 0.5:  # any order of picture types behind the leading I picture (B pictures need two anchors)
        types = [1, 2] + [int(x) for x in rng.choice([1, 2, 3], size=n)]
    seq = synth.generate_sequence(w, h, n, seed=seed, profile=profile, raw_fraction=raw, rgba=rgba, types=types)
    ref, dut = pyoracle.OracleStore(w, h, threads=4), abi.VideoStore(ctx, w, h)
    dut.set_tile_policy(policy)
    mirror = (seed & 3) == 0  # a quarter of the cases with the host mirror on (from the case's seed: the sequence of draws is as it was)
    try:
        run_and_compare(ref, Via(dut, form, mirror), seq, check_rgba=rgba)
    except AssertionError as e:
        raise AssertionError("MISMATCH: w=%d h=%d n=%d profile=%s raw=%.2f rgba=%d policy=%d form=%d mirror=%d seed=%d types=%s: %s" %
                             (w, h, n, profile, raw, rgba, policy, form, mirror, seed, types, e))
    finally:
        dut.close()
        ref.close()
    return len(seq), sum(len(s.mbs) for s in seq), policy, form


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    master = int(sys.argv[2]) if len(sys.argv) > 2 else 20260928
    from mpeg_amd import abi

    ctx = abi.Context(0)
    rng = np.random.default_rng(master)
    t0 = time.time()
    cases = pictures = mbs = 0
    by_policy = {0: 0, 1: 0, 2: 0}
    by_form = {0: 0, 1: 0, 2: 0}
    while time.time() - t0 < budget:
        try:
            p, m, policy, form = video_case(ctx, rng)
        except AssertionError as e:
            print(e)
            sys.exit(1)
        cases += 1
        pictures += p
        mbs += m
        by_policy[policy] += 1
        by_form[form] += 1
        if cases % 25 == 0:
            print("%d cases, %d pictures, %d macroblocks bit-exact (policy auto / int16 / int32: %d / %d / %d), %.0f s" % (
                cases, pictures, mbs, by_policy[0], by_policy[1], by_policy[2], time.time() - t0), flush=True)
    print("soak done: master seed %d, %d cases, %d pictures, %d macroblocks, every slot of every picture bit-exact vs the oracle "
          "(policy auto / int16 / int32: %d / %d / %d; hand-over units / sparse / device-packed: %d / %d / %d) in %.0f s" % (
              master, cases, pictures, mbs, by_policy[0], by_policy[1], by_policy[2], by_form[0], by_form[1], by_form[2], time.time() - t0))
    audio_soak(ctx, float(sys.argv[3]) if len(sys.argv) > 3 else budget / 4, master)
    ctx.close()


def audio_case(ctx, rng):
    """One seeded synthesis case (stream count, 1 - 3 calls on one state, frames per call, format, window arithmetic, sblimit, masked
    one-frame calls) against the oracle's synthesis, V ring state included; AssertionError on a mismatch.  -> stream-frames."""
    from mpeg_amd import abi, desc
    from oracle import pyoracle
    from parity import bits_equal
    frames = 0
    n_streams = int(rng.choice([1, 2, 3, 7, 33, 130, 600])) if rng.random() < 0.7 else int(rng.integers(1, 400))
    fma = int(rng.integers(0, 2))
    fmt = int(rng.choice([desc.AUDIO_F32N, desc.AUDIO_F32NLR, desc.AUDIO_F32, desc.AUDIO_S16]))
    calls = int(rng.integers(1, 4))
    ref, dut = pyoracle.OracleSynth(n_streams, fma), abi.AudioSynth(ctx, n_streams, fma)
    try:
        for _ in range(calls):
            n_frames = int(rng.integers(1, 40 if n_streams < 50 else 6))
            sblimit = int(rng.choice([8, 12, 27, 30, 32]))
            smp = rng.integers(-32768, 32768, size=(n_streams, n_frames, 2, 36, 32), dtype=np.int32)
            smp[..., sblimit:] = 0
            if rng.random() < 0.3 and n_frames == 1:  # (a masked call is one frame per stream: AudioBatch's tick)
                active = (rng.random(n_streams) < 0.7).astype(np.uint8)
                b = np.asarray(dut.synth_masked(smp, active, fmt))[active != 0]
                rows = []
                for i in np.nonzero(active)[0]:  # the oracle: the active streams one by one, each on its own state
                    one = pyoracle.OracleSynth(1, fma)
                    one.states = [ref.states[i]]
                    rows.append(one.synth(smp[i:i + 1], fmt)[0])
                a = np.stack(rows) if rows else b
            else:
                a, b = ref.synth(smp, fmt), dut.synth(smp, fmt)
            assert bits_equal(a, b), "AUDIO MISMATCH: streams=%d fma=%d fmt=%d frames=%d sblimit=%d" % (n_streams, fma, fmt, n_frames, sblimit)
            frames += n_streams * n_frames
        for st in sorted({0, n_streams - 1, int(rng.integers(0, n_streams))}):
            (va, pa), (vb, pb) = ref.get_state(st), dut.get_state(st)
            assert pa == pb and bits_equal(va, vb), "AUDIO STATE MISMATCH: streams=%d fma=%d stream %d" % (n_streams, fma, st)
    finally:
        dut.close()
    return frames


def audio_soak(ctx, budget, master):
    rng = np.random.default_rng(master + 1)
    t0 = time.time()
    cases = frames = 0
    while time.time() - t0 < budget:
        try:
            frames += audio_case(ctx, rng)
        except AssertionError as e:
            print(e)
            sys.exit(1)
        cases += 1
    print("audio soak done: master seed %d, %d cases, %d stream-frames (%d stereo sample pairs) bit-identical with the oracle's synthesis, "
          "V ring state included, in %.0f s" % (master + 1, cases, frames, frames * 1152, time.time() - t0))


if __name__ == "__main__":
    main()
