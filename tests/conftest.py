import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference (oracle/liboracle.so), built on demand."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def emu():
    """Test-only lane emulator of the GPU kernels (tests/kernel_emu)."""
    import emu as emu_mod
    emu_mod.lib()
    return emu_mod


@pytest.fixture(scope="session")
def hip_ctx():
    """A libmpeghip context on cuda:0.  Fails loudly (no fallback) without a GPU."""
    from mpeg_amd import abi
    ctx = abi.Context(0)
    yield ctx
    ctx.close()
