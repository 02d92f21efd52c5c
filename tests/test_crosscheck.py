"""The cross-stream checker itself (oracle/crosscheck.py), on the CPU with the lane emulator standing in for the device:
it passes a store whose streams are independent, and it FAILS a store in which one stream predicts from its neighbour's
frames — the error that identical content on every stream hides (every GPU bench leg and the full-size config-5 test
run this check)."""
import numpy as np

from mpeg_amd import desc, synth
from oracle import crosscheck


class _Batch:
    def __init__(self, store, sub, n_streams, steal=None):
        self.store, self.sub, self.n, self.steal = store, sub, n_streams, steal

    def run(self):
        s, n = self.sub, self.n
        if self.steal:  # the fault: stream `dst` predicts from stream `src`'s reference slots
            dst, src = self.steal
            for slot in {int(s.fwd), int(s.bwd)} - {int(s.cur)}:
                self.store.write_planes(dst, slot, *self.store.read_planes(src, slot))
        pics = np.repeat(s.pics, n)
        pics["stream"] = np.arange(n)
        pics["mb_first"] = np.arange(n) * len(s.mbs)
        mbs = np.tile(s.mbs, n)
        mbs["pic"] = np.repeat(np.arange(n), len(s.mbs))
        self.store.submit(pics, mbs, s.coefs)


def _run(emu, steal):
    w, h, n = 64, 48, 6
    seq = synth.generate_sequence(w, h, 7, seed=77)
    store = emu.EmuStore(w, h, n)
    batches = [_Batch(store, s, n, steal) for s in seq]
    for b in batches:
        b.run()
    return crosscheck.distinct_content_check(store, w, h, desc.geometry(w, h), n, seq, batches, probes=[0, 1, 3, 5])


def test_independent_streams_pass(oracle, emu):
    ok, text = _run(emu, None)
    assert ok, text
    assert "their own reference content" in text


def test_a_stream_that_reads_its_neighbours_frames_is_caught(oracle, emu):
    ok, text = _run(emu, (3, 2))
    assert not ok and "stream 3" in text, text


def test_probe_streams_span_the_store():
    assert crosscheck.probe_streams(1024) == [0, 1, 517, 1023]
    assert crosscheck.probe_streams(1) == [0]
    assert crosscheck.probe_streams(3) == [0, 1, 2]
