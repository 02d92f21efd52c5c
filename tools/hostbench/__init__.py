"""tools/hostbench — host-side measurement drivers over the C ABI (development aids, not product code).

staged_submit_rate(): pictures per second that `threads` host threads push through mpeghip_video_stage_* when
every device call carries one picture of `seq` (cycled) for each of `streams` streams: validation and packing into
the device format on the host, one H2D copy, reconstruction on the device."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
LIB = HERE / "libhostbench.so"


def build(force: bool = False) -> Path:
    src = HERE / "staged_rate.cpp"
    if not force and LIB.exists() and LIB.stat().st_mtime >= src.stat().st_mtime:
        return LIB
    libdir = ROOT / "mpeg_amd"
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", "-I", str(ROOT / "include"), str(src),
           "-o", str(LIB), "-L", str(libdir), "-lmpeghip", "-Wl,-rpath," + str(libdir)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hostbench build failed:\n" + r.stdout)
    return LIB


def staged_submit_rate(device: int, width: int, height: int, seq, streams: int, threads: int, seconds: float = 2.0,
                       verbose: bool = False, sparse: bool = True, device_pack: int = 0) -> float:
    """sparse: the pictures are handed over in the parser's own form (MPEGHIP_PIC_SPARSE: (position, level) pairs), what the
    product's parser emits; False: as 128-byte units.  device_pack: 0 = validated and packed by the putting threads, 1 = a
    device-packed stage (mpeghip_video_stage_begin_device: the puts copy, the device packs), 2 = the same with the pictures
    already in the pinned staging buffers (one picture of `seq` only): commit + PCIe + device, no host work per picture."""
    assert sparse or not device_pack
    from mpeg_amd import abi, desc
    abi.load_library()
    if sparse:
        import copy
        conv = []
        for s in seq:
            c = copy.copy(s)
            c.mbs, words = desc.to_sparse(s.mbs, s.coefs)
            c.coefs = words.view(np.uint8)
            c.pics = s.pics.copy()
            c.pics["flags"] |= desc.PIC_SPARSE
            conv.append(c)
        seq = conv
    H = C.CDLL(str(build()))
    H.hostbench_staged_submit_rate.restype = C.c_double
    H.hostbench_staged_submit_rate.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    n = len(seq)
    pics = np.ascontiguousarray(np.concatenate([s.pics[:1] for s in seq]))
    mbs = [np.ascontiguousarray(s.mbs) for s in seq]
    coefs = [np.ascontiguousarray(s.coefs).view(np.uint8).reshape(-1) for s in seq]
    mbs_p = (C.c_void_p * n)(*[m.ctypes.data for m in mbs])
    coefs_p = (C.c_void_p * n)(*[c.ctypes.data for c in coefs])
    n_mbs = np.array([len(m) for m in mbs], np.uint32)
    cbytes = np.array([c.nbytes for c in coefs], np.uint64)
    pps = H.hostbench_staged_submit_rate(device, width, height, streams, threads, seconds, n, pics.ctypes.data, mbs_p,
                                         n_mbs.ctypes.data, coefs_p, cbytes.ctypes.data, int(verbose), int(device_pack))
    if pps < 0:
        raise RuntimeError("hostbench_staged_submit_rate failed")
    return pps
