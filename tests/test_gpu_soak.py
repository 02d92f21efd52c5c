"""-m gpu: a slice of the randomised differential soak (tools/gpu_soak.py — whose full runs, 8 - 11 thousand cases per seed, are under
profiles/): seeded cases over geometry (16 x 16 .. 420 x 300, one in twelve up to 1920 x 1088) x 2 - 8 pictures in GOP order or a
random order of picture types x typical / dense content x 0 / 5 / 20 % int32 snapshot blocks x fused Frame.RGBA x kernel policy (the
wide kernel, both recon_kernel instances) x hand-over form (units, sparse words packed on the host, sparse words packed on the
device) — every slot of every picture against the oracle through the C ABI — and its audio phase (stream counts, calls in a row on
one state, the four formats, both window arithmetics, masked calls, the V ring state).  480 + 48 cases, fixed seeds."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools import gpu_soak

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [6001, 6002, 6003, 6004, 6005, 6006, 6007, 6008])
def test_video_soak_slice(oracle, hip_ctx, seed):
    rng = np.random.default_rng(seed)
    seen_policy, seen_form, pictures = set(), set(), 0
    for _ in range(60):
        p, _, policy, form = gpu_soak.video_case(hip_ctx, rng)
        pictures += p
        seen_policy.add(policy)
        seen_form.add(form)
    assert seen_policy == {0, 1, 2} and seen_form == {0, 1, 2} and pictures >= 150


@pytest.mark.parametrize("seed", [7001, 7002, 7003])
def test_audio_soak_slice(oracle, hip_ctx, seed):
    rng = np.random.default_rng(seed)
    assert sum(gpu_soak.audio_case(hip_ctx, rng) for _ in range(16)) > 0
