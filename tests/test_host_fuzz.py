"""Mutated streams: the product's parser (mpeg_amd/host) against the oracle's decoder on streams with random bytes and bits
changed — frame by frame / sample by sample, bit-exact.  The reference's golden stream pins a handful of damaged
macroblocks; this walks the error paths at large (invalid blocks, stale blockData snapshots, macroblocks addressed twice,
vectors out of range, broken headers, truncated frames).

Where the reference would PANIC (a copyMacroblock source slice out of range, video_noasm.go:48-50) there is nothing to
restate; product and oracle define the same thing: the whole macroblock is dropped (mpeg_amd/host/video.cpp:
emitPrediction / endMacroblockRecord; oracle/mpeg_oracle.c: predict_macroblock)."""
import numpy as np
import pytest

import hostlib


def mutate(data: bytes, rng, first: int, last: int) -> bytes:
    d = bytearray(data)
    for _ in range(int(rng.integers(3, 40))):
        p = int(rng.integers(first, last))
        mode = int(rng.integers(0, 3))
        if mode == 0:
            d[p] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            d[p] = int(rng.integers(0, 256))
        else:
            d[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
    return bytes(d)


def compare_video(oracle, data, frames):
    ref, dut = oracle.VideoDecoder(data), hostlib.HostVideo(data, emu_flavour=0)
    try:
        for i in range(frames):
            a, b = ref.decode(), dut.decode()
            assert (a is None) == (b is None), "frame %d: one decoder has ended" % i
            if a is None:
                break
            for pa, pb in zip(oracle.frame_planes(a), hostlib.frame_planes(b)):
                assert np.array_equal(pa, pb), "frame %d" % i
        return dut.stats()
    finally:
        ref.close()
        dut.close()


@pytest.mark.parametrize("seed", range(8))
def test_mutated_damaged_stream(oracle, golden_dir, seed):
    data = (golden_dir / "test.mpeg1video").read_bytes()
    rng = np.random.default_rng(1000 + seed)
    seen = {"range_skips": 0, "invalid_blocks": 0, "duplicate_splits": 0, "raw_macroblocks": 0}
    for _ in range(8):
        st = compare_video(oracle, mutate(data, rng, 200, 60000), 70)     # the first 70 frames live in the first ~60 kB
        for k in seen:
            seen[k] += st[k]
    assert seen["invalid_blocks"] and seen["raw_macroblocks"] and seen["duplicate_splits"]


def test_mutations_reach_the_out_of_range_vectors(oracle, golden_dir):
    """the case the reference cannot answer (it panics): seeds known to produce vectors out of range"""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    rng = np.random.default_rng(2)
    skips = 0
    for _ in range(30):
        skips += compare_video(oracle, mutate(data, rng, 200, 60000), 70)["range_skips"]
    assert skips > 0


@pytest.mark.parametrize("seed", range(4))
def test_mutated_clean_stream(oracle, golden_dir, seed):
    es, _ = oracle.ps_extract((golden_dir / "test.mpg").read_bytes(), 0xE0)
    rng = np.random.default_rng(2000 + seed)
    for _ in range(8):
        compare_video(oracle, mutate(es, rng, 200, 70000), 70)


@pytest.mark.parametrize("seed", range(3))
def test_mutated_audio_stream(oracle, emu, golden_dir, seed):
    data = (golden_dir / "test.mp2").read_bytes()
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    rng = np.random.default_rng(3000 + seed)
    for _ in range(12):
        d = mutate(data, rng, 4, len(data) - 8)
        ref, dut = oracle.AudioDecoder(d, 0), hostlib.HostAudio(d, fma=0, fmt=0, window=win)
        try:
            n = 0
            while True:
                a, b = ref.decode(), dut.decode()
                assert (a is None) == (b is None), "frame %d: one decoder has ended" % n
                if a is None:
                    break
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "frame %d" % n
                n += 1
        finally:
            ref.close()
            dut.close()


def test_mutated_program_streams_through_the_facade(emu, golden_dir):
    """Robustness of the whole host stack (Demux, MPEG facade, both parsers) on damaged program streams: no crash, no
    hang, sane counts.  (The second stream of this seed once corrupted the heap: a read past the end of the buffer — where
    the reference panics, buffer.go:246-255 — left the bit index behind the end, and the next discardReadBytes erased a
    negative range.  tools/tsan/fuzz_facade.cpp and fuzz_streams.cpp run hundreds of such streams under ASan + UBSan.)"""
    data0 = (golden_dir / "test.mpg").read_bytes()
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    rng = np.random.default_rng(1)
    for it in range(12):
        d = bytearray(data0)
        for _ in range(int(rng.integers(3, 60))):
            p = int(rng.integers(0, len(d) - 8))
            mode = int(rng.integers(0, 4))
            if mode == 0:
                d[p] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                d[p] = int(rng.integers(0, 256))
            elif mode == 2:
                d[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
            else:
                d[p:p + 4] = b"\x00\x00\x01" + bytes([int(rng.choice([0xBA, 0xBB, 0xE0, 0xC0, 0xB3, 0x00, 0xB9]))])
        if rng.integers(0, 4) == 0:
            d = d[:int(rng.integers(1000, len(d)))]
        try:
            m = hostlib.HostMpeg(bytes(d), window=win)
        except RuntimeError:
            continue                                  # (a stream the constructor refuses, as mpeg.New does: ErrInvalidMPEG)
        nv = na = 0
        while nv < 400 and m.decode_video() is not None:
            nv += 1
        while na < 400 and m.decode_audio() is not None:
            na += 1
        m.seek(float(rng.uniform(0, 9)), int(rng.integers(0, 2)))
        for _ in range(10):
            if m.decode_video() is None:
                break
        m.seek_frame(float(rng.uniform(0, 9)), 1)
        assert 0 <= m.duration < 60
        m.rewind()
        m.decode(0.5)
        m.close()
        assert nv <= 400 and na <= 400


def test_mutated_streams_through_the_batch(oracle, golden_dir):
    """mpeg::VideoBatch (4 parse threads, staged hand-over, one device call per tick) on six differently damaged streams:
    every stream's every frame equals the oracle's decode of that stream."""
    data = (golden_dir / "test.mpeg1video").read_bytes()
    rng = np.random.default_rng(1)
    for _ in range(2):
        streams = [mutate(data, rng, 200, 60000) for _ in range(6)]
        batch = hostlib.HostBatch(len(streams), threads=4)
        refs = [oracle.VideoDecoder(s) for s in streams]
        try:
            for s in streams:
                batch.add_stream(s)
            for tick in range(50):
                batch.decode_all(True)
                for i, ref in enumerate(refs):
                    a, f = ref.decode(), batch.frame(i)
                    assert (a is None) == (f is None), "tick %d stream %d" % (tick, i)
                    if a is not None:
                        for pa, pb in zip(oracle.frame_planes(a), hostlib.frame_planes(f)):
                            assert np.array_equal(pa, pb), "tick %d stream %d" % (tick, i)
        finally:
            for r in refs:
                r.close()
            batch.close()


def test_mutated_program_streams_demux(oracle, golden_dir):
    """mpeg::Demux against the oracle's packet extractor (the subset of demux.go:473-584 it restates) on damaged program
    streams: the same video and audio payload bytes, packet for packet."""
    data0 = (golden_dir / "test.mpg").read_bytes()
    rng = np.random.default_rng(7)
    for it in range(25):
        d = bytearray(data0)
        for _ in range(int(rng.integers(1, 30))):
            p = int(rng.integers(0, len(d) - 8))
            mode = int(rng.integers(0, 4))
            if mode == 0:
                d[p] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                d[p] = int(rng.integers(0, 256))
            elif mode == 2:
                d[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
            else:
                d[p:p + 4] = b"\x00\x00\x01" + bytes([int(rng.choice([0xBA, 0xBB, 0xE0, 0xC0, 0xB3, 0x00, 0xB9]))])
        d = bytes(d)
        try:
            dm = hostlib.HostDemux(d)
        except RuntimeError:
            continue
        got = {hostlib.PACKET_VIDEO_1: [], hostlib.PACKET_AUDIO_1: []}
        for _ in range(5000):
            pk = dm.decode()
            if pk is None:
                break
            if pk[0] in got:
                got[pk[0]].append(pk[2])
        dm.close()
        for typ, parts in got.items():
            want, n_packets = oracle.ps_extract(d, typ)
            assert b"".join(parts) == want and len(parts) == n_packets, "stream %d, packets of type %#x" % (it, typ)


def test_mutated_audio_streams_through_the_batch(oracle, emu, golden_dir):
    """mpeg::AudioBatch (one synthesis call per tick for all streams) on five differently damaged MP2 streams: every
    stream's every sample block equals the oracle's decode of that stream, bit for bit."""
    data = (golden_dir / "test.mp2").read_bytes()
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    rng = np.random.default_rng(11)
    streams = [mutate(data, rng, 4, len(data) - 8) for _ in range(5)]
    batch = hostlib.HostAudioBatch(len(streams), fmt=0, window=win)
    refs = [oracle.AudioDecoder(s, 0) for s in streams]
    try:
        for s in streams:
            batch.add_stream(s)
        for tick in range(400):
            produced = batch.decode_all()
            for i, ref in enumerate(refs):
                a, s = ref.decode(), batch.samples(i)
                assert (a is None) == (s is None), "tick %d stream %d" % (tick, i)
                if a is not None:
                    assert np.array_equal(a.view(np.uint32), s.view(np.uint32)), "tick %d stream %d" % (tick, i)
            if produced == 0:
                break
        assert tick > 300
    finally:
        for r in refs:
            r.close()
        batch.close()


def test_mutated_written_streams_of_random_sizes(oracle):
    """Streams of random picture sizes (written from synthetic descriptors, tests/mpeg1_writer.py), clean and damaged:
    larger and odd-sized pictures, typical and dense content, through the product's parser against the oracle's."""
    import mpeg1_writer
    from mpeg_amd import synth
    rng = np.random.default_rng(2)
    for _ in range(8):
        w, h = int(rng.integers(16, 260)), int(rng.integers(16, 200))
        seq = synth.generate_sequence(w, h, int(rng.integers(3, 8)), seed=int(rng.integers(1, 1 << 30)),
                                      profile=str(rng.choice(["typical", "dense"])))
        es = mpeg1_writer.write_sequence(w, h, seq)
        compare_video(oracle, es, 20)
        for _ in range(2):
            compare_video(oracle, mutate(es, rng, 12, len(es) - 8), 20)
