"""In-tree builds.  Everything is compiled with explicit command lines (no cmake,
no JIT cache) so that the resulting .so files sit next to their sources and
travel with the repository snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "mpeg_amd" / "csrc"
HOST = ROOT / "mpeg_amd" / "host"
INCLUDE = ROOT / "include"

LIBMPEGHIP = ROOT / "mpeg_amd" / "libmpeghip.so"
LIBMPEGHOST = ROOT / "mpeg_amd" / "libmpeghost.so"

# -fno-slp-vectorize: packed f32 math (v_pk_mul/add_f32) issues at half rate on gfx950, so pairing two scalar
# operations gains nothing and costs the register shuffles (audio kernel: 124 -> 97 VGPRs, +4 %)
# -amdgpu-kernarg-preload-count=14: the leading scalar kernel arguments arrive in SGPRs with the wave (gfx940+; the compiler keeps
# a load-them-yourself entry for firmware without the feature) — recon_kernel's first 14 dwords are what a wave needs before
# its chunk header is back (mpeghip.hip)
# -structurizecfg-skip-uniform-regions: the structuriser leaves regions whose branches are all wave-uniform as they are.  Without
# it (this compiler's default) recon_kernel's per-macroblock mode dispatch — uniform by construction — is linearised like divergent
# control flow: boolean flags in SGPR pairs, `s_andn2_b64 vcc, exec, flag; s_cbranch_vccnz` in the place of `s_cmp; s_cbranch_scc`.
# Same sources, interleaved on one box (profiles/round5_a_ab_structurizer_skips_uniform_regions.txt): scalar instructions per wave
# 349 -> 319, typical 0.621 -> 0.633 of the roofline, dense 0.550 -> 0.561, bit-exact.
# -amdgpu-sched-strategy=max-ilp: the machine scheduler orders for instruction-level parallelism first (its default weighs register
# pressure first).  With the round-5 kernel: typical +0.3 ... 0.6 % in six of six interleaved rounds on two boxes, dense +0.1 ... 0.4 %,
# audio and the one-picture launch within noise (profiles/round5_j_ab_scheduler_strategies.txt); register counts rise (the int16-tile instance 49 -> 57,
# the audio kernel 70 -> 80) without costing a resident wave.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm",
               "-amdgpu-kernarg-preload-count=14", "-mllvm", "-structurizecfg-skip-uniform-regions", "-mllvm",
               "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared"]


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def _run(cmd, cwd=None):
    r = subprocess.run([str(c) for c in cmd], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(str(c) for c in cmd), r.stdout))
    return r.stdout


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmpeghip is HIP-only and cannot be built without ROCm")


def build_libmpeghip(force: bool = False) -> Path:
    """hipcc cross-compiles the gfx950 code objects; no GPU is needed to build."""
    srcs = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hip")) + [INCLUDE / "mpeghip.h"]
    if not force and _newer(LIBMPEGHIP, srcs):
        return LIBMPEGHIP
    _run([hipcc_path(), *HIPCC_FLAGS, "-I", INCLUDE, "-I", CSRC, CSRC / "mpeghip.hip", "-o", LIBMPEGHIP])
    return LIBMPEGHIP


def build_libmpeghost(force: bool = False) -> Path:
    """Host-side mirror of the reference API (bitstream parse -> descriptors).
    Plain C++; it dlopens nothing and links libmpeghip for all device work."""
    srcs = sorted(HOST.glob("*.h")) + sorted(HOST.glob("*.hpp")) + sorted(HOST.glob("*.cpp")) + [INCLUDE / "mpeghip.h"]
    cpps = sorted(HOST.glob("*.cpp"))
    if not cpps:
        raise RuntimeError("no host sources")
    if not force and _newer(LIBMPEGHOST, srcs + [LIBMPEGHIP]):
        return LIBMPEGHOST
    build_libmpeghip()
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", "-I", INCLUDE, "-I", HOST, *cpps,
          "-o", LIBMPEGHOST, "-L", LIBMPEGHIP.parent, "-lmpeghip", "-Wl,-rpath,$ORIGIN"])
    return LIBMPEGHOST


def build_all(force: bool = False):
    out = [build_libmpeghip(force)]
    if sorted(HOST.glob("*.cpp")):
        out.append(build_libmpeghost(force))
    return out
