"""-m gpu: MP2 synthesis kernel through the C ABI vs the oracle.  The bar of the
north star is RMS <= 1e-6; the kernel is built to be bit-identical (no-FMA and
window-FMA variants), so bit equality is asserted and RMS reported on top."""
import numpy as np
import pytest

from mpeg_amd import abi, desc, synth
from parity import bits_equal, mirror_ring

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fma", [desc.AUDIO_FMA_NONE, desc.AUDIO_FMA_WINDOW])
@pytest.mark.parametrize("fmt", [desc.AUDIO_F32N, desc.AUDIO_F32NLR, desc.AUDIO_F32, desc.AUDIO_S16])
def test_formats_and_state(oracle, hip_ctx, fma, fmt):
    s = synth.audio_frames(3, 5)
    ref, dut = oracle.OracleSynth(3, fma), abi.AudioSynth(hip_ctx, 3, fma)
    for _ in range(3):
        assert bits_equal(ref.synth(s, fmt), dut.synth(s, fmt))
        for st in range(3):
            (va, pa), (vb, pb) = ref.get_state(st), dut.get_state(st)
            assert pa == pb and bits_equal(va, vb)
    dut.close()


@pytest.mark.parametrize("fma,want", [(0, 0xf1b76cdf8e6cdea5), (1, 0x50f3ab75f5fb0fb5)])
def test_golden_hash_through_the_gpu(oracle, hip_ctx, golden_dir, fma, want):
    """test.mp2's real sub-band samples -> HIP synthesis -> the reference's golden FNV (mpeg_test.go:193-197)."""
    dec = oracle.AudioDecoder((golden_dir / "test.mp2").read_bytes(), fma)
    frames = []
    while True:
        r = dec.decode(True)
        if r is None:
            break
        frames.append(r[1])
    S = np.stack(frames)[None]
    dut, h, i = abi.AudioSynth(hip_ctx, 1, fma), oracle.FNV_OFFSET, 0
    for chunk in (1, 2, 5, 17, 100, 1000):
        part = S[:, i:i + chunk]
        if part.shape[1]:
            h = oracle.fnv1a64(dut.synth(part, desc.AUDIO_F32N), h)
        i += chunk
    dut.close()
    assert h == want


def test_config4_256_streams(oracle, hip_ctx):
    """BASELINE config 4: 256 stereo streams; 100 consecutive frames so the V ring wraps many times."""
    n_streams, n_frames = 256, 100
    s = synth.audio_frames(n_streams, n_frames)
    dut = abi.AudioSynth(hip_ctx, n_streams, desc.AUDIO_FMA_NONE)
    got = dut.synth(s, desc.AUDIO_F32N)
    ref = oracle.OracleSynth(n_streams, 0)
    pick = [0, 1, 17, 128, 255]
    want = oracle.OracleSynth(len(pick), 0).synth(s[pick], desc.AUDIO_F32N)
    rms = float(np.sqrt(np.mean((got[pick].astype(np.float64) - want.astype(np.float64)) ** 2)))
    assert rms <= 1e-6, rms          # the north star's tolerance
    assert bits_equal(got[pick], want)  # and the stronger property this kernel is built for
    dut.close()


@pytest.mark.parametrize("n_streams,n_frames", [(5, 33), (300, 16), (1100, 9), (3, 2)])
def test_time_slicing_is_bit_identical(oracle, hip_ctx, n_streams, n_frames):
    """A launch is split along time so that the GPU is full (slices = workgroups per CU x CUs / streams, at least 4 frames
    each; the history is rebuilt from the samples): 8, 4, 1 and 1 slices here."""
    s = synth.audio_frames(n_streams, n_frames)
    ref, dut = oracle.OracleSynth(n_streams, 0), abi.AudioSynth(hip_ctx, n_streams, 0)
    for _ in range(2):
        assert bits_equal(ref.synth(s), dut.synth(s))
        (va, pa), (vb, pb) = ref.get_state(n_streams - 1), dut.get_state(n_streams - 1)
        assert pa == pb and bits_equal(va, vb)
    dut.close()


def test_zero_frames_and_state_transplant(oracle, hip_ctx):
    s = synth.audio_frames(1, 4)
    src = abi.AudioSynth(hip_ctx, 1)
    src.synth(s[:, :3])
    v, p = src.get_state(0)          # a genuine Audio.v / vPos
    dut = abi.AudioSynth(hip_ctx, 1)
    dut.set_state(0, v, p)
    out = dut.synth(np.zeros((1, 0, 2, 36, 32), np.int32))
    assert out.shape == (1, 0, 2304)
    v2, p2 = dut.get_state(0)
    assert p2 == p and bits_equal(v, v2)
    # the transplanted stream continues exactly like the original (and like the oracle)
    ref = oracle.OracleSynth(1, 0)
    ref.synth(s[:, :3], desc.AUDIO_F32N)
    want = ref.synth(s[:, 3:], desc.AUDIO_F32N)
    assert bits_equal(src.synth(s[:, 3:]), want) and bits_equal(dut.synth(s[:, 3:]), want)
    src.close()
    dut.close()


def test_set_state_refuses_a_ring_that_is_not_a_synthesis_state(hip_ctx):
    """Audio.v slots are signed mirrors of 32 DCT outputs; anything else cannot come from the reference."""
    dut = abi.AudioSynth(hip_ctx, 1)
    with pytest.raises(abi.MpegHipError) as ei:
        dut.set_state(0, np.arange(2048, dtype=np.float32).reshape(2, 1024), 192)
    assert ei.value.code == abi.ERR_INVALID
    dut.set_state(0, np.zeros((2, 1024), np.float32), 192)  # zeros (a fresh decoder) are fine
    for bad in (-64, 1024, 100):
        with pytest.raises(abi.MpegHipError):
            dut.set_state(0, None, bad)
    dut.close()


@pytest.mark.parametrize("scale", [2.0 ** -125, 2.0 ** -140, 1.0])
def test_scaling_of_tiny_sums(oracle, hip_ctx, scale):
    """Outputs whose window sums fall inside 2^-119..2^-95 must take the kernel's IEEE division (the
    short form is only proven outside, tests/proofs/div_const.c)."""
    rng = np.random.default_rng(5)
    v = mirror_ring(rng.integers(-999, 1000, (2, 16, 32)).astype(np.float32) * np.float32(scale))
    ref, dut = oracle.OracleSynth(1, 0), abi.AudioSynth(hip_ctx, 1)
    ref.set_state(0, v, 320)
    dut.set_state(0, v, 320)
    s = np.zeros((1, 1, 2, 36, 32), np.int32)
    a, b = ref.synth(s, desc.AUDIO_F32N), dut.synth(s, desc.AUDIO_F32N)
    assert bits_equal(a, b) and np.count_nonzero(a) > 500
    dut.close()


def test_masked_synth_leaves_idle_streams_alone(oracle, hip_ctx):
    """mpeghip_audio_synth_masked: streams with active == 0 keep their V ring / vPos and their output rows;
    the others behave exactly like mpeghip_audio_synth (what AudioBatch needs when a stream has no frame)."""
    n, rng = 6, np.random.default_rng(9)
    s = synth.audio_frames(n, 7)
    ref = [oracle.OracleSynth(1, 0) for _ in range(n)]
    dut = abi.AudioSynth(hip_ctx, n)
    for step in range(5):
        mask = (rng.random(n) < 0.6).astype(np.uint8)
        if step == 2:
            mask[:] = 0  # nobody: the call must be a no-op
        frames = s[:, step:step + 1]
        out = dut.synth_masked(frames, mask, out=np.full((n, 1, 2304), 7.0, np.float32))
        for i in range(n):
            if mask[i]:
                assert bits_equal(out[i], ref[i].synth(frames[i:i + 1], desc.AUDIO_F32N)[0])
            else:
                assert (out[i] == 7.0).all()
            (va, pa), (vb, pb) = ref[i].get_state(0), dut.get_state(i)
            assert pa == pb and bits_equal(va, vb)
    dut.close()
