"""Python face of tests/kernel_emu (TEST INFRASTRUCTURE): runs the lane functions of
the GPU kernels on the CPU so their index logic can be compared with the oracle
without a GPU.  Same surface as mpeg_amd.abi.VideoStore / AudioSynth."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from mpeg_amd import desc, synth

HERE = Path(__file__).resolve().parent / "kernel_emu"
ROOT = HERE.parent.parent
LIB = HERE / "libkernel_emu.so"

_lib = None
_libs = {}      # variant tag -> loaded library ("" = the product's build options)
_variant = ("", ())


def select(tag="", flags=()):
    """Switch the emulator to a build of the lane headers with other compile-time options (-D...); select() goes back to
    the product's."""
    global _lib, _variant
    _variant = (tag, tuple(flags))
    _lib = _libs.get(tag)


def build(force=False):
    tag, flags = _variant
    out = LIB if not tag else HERE / ("libkernel_emu_%s.so" % tag)
    srcs = [HERE / "emu.cpp"] + sorted((ROOT / "mpeg_amd" / "csrc").glob("*.h")) + [ROOT / "include" / "mpeghip.h"]
    if not force and out.exists() and all(s.stat().st_mtime <= out.stat().st_mtime for s in srcs):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DMPG_EMU_CHECKS", "-Wall", "-Wextra",
           "-Wno-unknown-pragmas", *flags, "-I", str(ROOT / "include"), "-I", str(ROOT / "mpeg_amd" / "csrc"),
           str(HERE / "emu.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulator build failed:\n" + r.stdout)
    return out


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(str(build()))
        P = C.c_void_p
        L.emu_video_run.restype = C.c_int
        L.emu_video_run.argtypes = [P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_uint32, P, C.c_uint32, P, P, P, C.c_uint64]
        L.emu_video_run_sparse.restype = C.c_int
        L.emu_video_run_sparse.argtypes = [P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_uint32, P, C.c_uint32, P,
                                           C.c_uint64, P, P, C.c_uint64]
        L.emu_set_tile_policy.restype = None
        L.emu_set_tile_policy.argtypes = [C.c_int]
        L.emu_make_qtable.restype = None
        L.emu_make_qtable.argtypes = [P, P, P]
        for f in (L.emu_pack, L.emu_pack_narrow):
            f.restype = C.c_uint32
            f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, P, P, P, P, P, P]
        L.emu_set_device_pack.restype = None
        L.emu_set_device_pack.argtypes = [C.c_int]
        L.emu_set_pack_window.restype = None
        L.emu_set_pack_window.argtypes = [C.c_uint32]
        L.emu_pack_sparse.restype = C.c_uint32
        L.emu_pack_sparse.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, P, P, P, C.c_uint64, P, P, P, C.c_uint64]
        L.emu_pack_device_picture.restype = C.c_uint64
        L.emu_pack_device_picture.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, P, C.c_uint32, C.c_uint32, C.c_uint32, P, P, P, P, P]
        L.emu_host_has_avx512.restype = C.c_int
        L.emu_audio_slice_range.restype = None
        L.emu_audio_slice_range.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, P, P]
        L.emu_relayout.restype = None
        L.emu_relayout.argtypes = [P, P, C.c_uint32, C.c_uint32, C.c_int]
        L.emu_rgba_convert.restype = None
        L.emu_rgba_convert.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P]
        L.emu_audio_run.restype = C.c_int
        L.emu_audio_run.argtypes = [P, P, P, P, P, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32]
        for n in ("emu_avg4", "emu_avg2", "emu_xcd_chunk", "emu_ycbcr"):
            getattr(L, n).restype = C.c_uint32
        L.emu_avg4.argtypes = [C.c_uint32] * 4
        L.emu_avg2.argtypes = [C.c_uint32] * 2
        L.emu_xcd_chunk.argtypes = [C.c_uint32] * 2
        L.emu_ycbcr.argtypes = [C.c_uint32] * 3
        _lib = _libs[_variant[0]] = L
    return _lib


def set_tile_policy(policy=0):
    """Which instance of the reconstruction kernel the emulator runs (mpeghip_video_set_tile_policy): 0 = the library's
    per-batch rule, 1 = the instance that transposes across lanes, 2 = the instance for dense units (transposition through LDS)."""
    lib().emu_set_tile_policy(policy)


def set_wide(on=0):
    """1: every chunk runs as recon_wide_kernel runs it (four waves per chunk, two workgroup barriers) instead of as recon_kernel's
    one wave does."""
    lib().emu_set_wide(int(on))


def set_device_pack(on=0):
    """1: sparse pictures go through the DEVICE packer's lane functions (video_pack_lane.h: what pack_kernel runs) instead of the
    host packer."""
    lib().emu_set_device_pack(int(on))


def set_pack_window(dwords=4096):
    """The device packer's LDS window in dwords (pack_kernel's is 4096): tests shrink it so that a wave's reads fall inside,
    across and beyond it."""
    lib().emu_set_pack_window(int(dwords))


CHUNK_DWORDS = 32   # video_recon_lane.h: kRcChunkDwords (8 header dwords + 4 records of 6)


def pack_sparse_host(g, stride, rgba_stride, pic, mbs, words, out_room=None):
    """The host packer on one sparse picture -> (chunks [n, 32], words) or None if it refuses the picture."""
    pic = np.ascontiguousarray(np.asarray(pic).reshape(1), desc.PIC_DTYPE)
    mbs = np.ascontiguousarray(mbs, desc.MB_DTYPE)
    words = np.ascontiguousarray(words, np.uint32)
    n_chunks = (int(pic["mb_count"][0]) + 3) // 4
    chunks = np.zeros((n_chunks + 1, CHUNK_DWORDS), np.uint32)
    out = np.full(len(words) + 64 + 16, 0xDEADBEEF, np.uint32)
    nw = C.c_uint32(0)
    room = len(words) + 64 if out_room is None else out_room
    n = lib().emu_pack_sparse(g["luma_w"], g["luma_h"], stride, rgba_stride, _ptr(pic), _ptr(mbs), _ptr(words), len(words), _ptr(chunks),
                              _ptr(out), C.byref(nw), room)
    if n == 0xffffffff:
        return None
    assert n == n_chunks
    return chunks[:n], out[:nw.value]


def pack_sparse_device(g, stride, rgba_stride, pic, mbs, words, word_first=0, chunk_first=0):
    """The device packer's lane functions on one sparse picture -> (error word, chunks [n, 32], words array as long as the input
    + word_first, use bits)."""
    pic = np.ascontiguousarray(np.asarray(pic).reshape(1), desc.PIC_DTYPE)
    mbs = np.ascontiguousarray(mbs, desc.MB_DTYPE)
    words = np.ascontiguousarray(words, np.uint32)
    n_chunks = (int(pic["mb_count"][0]) + 3) // 4
    chunks = np.zeros((chunk_first + n_chunks + 1, CHUNK_DWORDS), np.uint32)
    staged = np.concatenate([np.full(word_first, 0xABABABAB, np.uint32), words, np.full(16, 0xABABABAB, np.uint32)])  # (the buffer is padded)
    out = np.full(word_first + len(words) + 1, 0xDEADBEEF, np.uint32)
    use = C.c_uint32(0)
    err = lib().emu_pack_device_picture(g["luma_w"], g["luma_h"], stride, rgba_stride, _ptr(pic), word_first, len(words), chunk_first,
                                        _ptr(mbs), _ptr(staged), _ptr(chunks), _ptr(out), C.byref(use))
    assert out[-1] == 0xDEADBEEF and (chunks[-1] == 0).all(), "the packer wrote past its picture"
    return err, chunks[chunk_first:chunk_first + n_chunks], out[:-1], use.value


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _qtable(intra, non_intra):
    """One stream's dequantisation table in the device layout (video_recon_lane.h: rc_make_qtable)."""
    t = np.zeros(256, np.uint8)
    i, n = np.ascontiguousarray(intra, np.uint8), np.ascontiguousarray(non_intra, np.uint8)
    lib().emu_make_qtable(_ptr(t), _ptr(i), _ptr(n))
    return t


class EmuStore:
    def __init__(self, width, height, n_streams=1, guard=False):
        """guard: the frame store ends exactly at an inaccessible page (mmap + mprotect), sized as mpeghip_video_open sizes it
        (slots + the tail pad behind the last one): a lane that reads further than the product's allocation faults
        (checked once with the pad taken out: SIGSEGV in the gather)."""
        self.g = desc.geometry(width, height)
        self.n_streams = n_streams
        self.stride = (self.g["frame_bytes"] + 64 + 255) // 256 * 256
        self.tail_pad = (self.g["luma_w"] + 64 + 255) // 256 * 256           # mpeghip.hip: mpeghip_video_open
        total = self.stride * 3 * n_streams + self.tail_pad
        if guard:
            import mmap
            page = mmap.PAGESIZE
            room = (total + page - 1) // page * page
            self._map = mmap.mmap(-1, room + page)
            base = C.addressof(C.c_char.from_buffer(self._map))
            libc = C.CDLL(None, use_errno=True)
            libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
            assert libc.mprotect(base + room, page, 0) == 0                   # PROT_NONE
            whole = np.frombuffer(self._map, np.uint8, room)
            self.frames = whole[room - total:]                                # ends exactly at the guard page
            self.frames[:] = 0
        else:
            self.frames = np.zeros(total, np.uint8)
        self.rgba_stride = (width * height * 4 + 255) // 256 * 256
        self.rgba = np.zeros(self.rgba_stride * 3 * n_streams, np.uint8)
        self.qmat = np.zeros((n_streams + 4, 256), np.uint8)   # (+ the padding the lanes may read, as on the device)
        for s in range(n_streams):
            self.qmat[s] = _qtable(synth.INTRA_Q, synth.NON_INTRA_Q)
        self._rgba_init = False

    def set_quant(self, stream, intra, non_intra):
        self.qmat[stream] = _qtable(intra, non_intra)

    def _slot(self, stream, slot):
        o = (stream * 3 + slot) * self.stride
        return self.frames[o:o + self.stride]

    def submit(self, pics, mbs, coefs):
        pics = np.ascontiguousarray(pics, desc.PIC_DTYPE)
        mbs = np.ascontiguousarray(mbs, desc.MB_DTYPE)
        coefs = np.ascontiguousarray(coefs).view(np.uint8).reshape(-1)
        if (pics["flags"] & desc.PIC_RGBA).any() and not self._rgba_init:
            for s in range(self.n_streams):
                for slot in range(3):
                    self.rgba_convert(slot, s, 1)
            self._rgba_init = True
        g = self.g
        rc = lib().emu_video_run(_ptr(self.frames), self.stride, g["luma_w"], g["luma_h"], g["width"], g["height"],
                                 _ptr(pics), len(pics), _ptr(mbs), len(mbs), _ptr(coefs),
                                 _ptr(self.qmat), _ptr(self.rgba), self.rgba_stride)
        assert rc == 0

    def submit_sparse(self, pic, mbs, words):
        """One picture in the sparse hand-over form (desc.to_sparse): the library's sparse packer, then the kernel's lanes."""
        pics = np.ascontiguousarray(np.asarray(pic).reshape(1), desc.PIC_DTYPE)
        mbs = np.ascontiguousarray(mbs, desc.MB_DTYPE)
        words = np.ascontiguousarray(words, np.uint32)
        if (pics["flags"] & desc.PIC_RGBA).any() and not self._rgba_init:
            for s in range(self.n_streams):
                for slot in range(3):
                    self.rgba_convert(slot, s, 1)
            self._rgba_init = True
        g = self.g
        rc = lib().emu_video_run_sparse(_ptr(self.frames), self.stride, g["luma_w"], g["luma_h"], g["width"], g["height"],
                                        _ptr(pics), 1, _ptr(mbs), len(mbs), _ptr(words), len(words),
                                        _ptr(self.qmat), _ptr(self.rgba), self.rgba_stride)
        return rc

    def read_planes(self, stream, slot):
        f, g = self._slot(stream, slot), self.g
        L, Cb = g["luma_bytes"], g["chroma_bytes"]
        lin = np.zeros(L + 2 * Cb, np.uint8)    # the frame store is tiled (video_lane.h): untile, as read_planes does
        lib().emu_relayout(_ptr(f), _ptr(lin), g["luma_w"], g["luma_h"], 1)
        return lin[:L].copy(), lin[L:L + Cb].copy(), lin[L + Cb:L + 2 * Cb].copy()

    def write_planes(self, stream, slot, y, cb, cr, pad=None):
        f, g = self._slot(stream, slot), self.g
        L, Cb = g["luma_bytes"], g["chroma_bytes"]
        lin = np.concatenate([np.asarray(y, np.uint8).reshape(-1), np.asarray(cb, np.uint8).reshape(-1), np.asarray(cr, np.uint8).reshape(-1)])
        lib().emu_relayout(_ptr(f), _ptr(lin), g["luma_w"], g["luma_h"], 0)
        if pad is not None:
            f[L + 2 * Cb:L + 2 * Cb + g["luma_w"] * 16] = pad

    def rgba_convert(self, slot, stream0=0, n=None):
        g = self.g
        for s in range(stream0, stream0 + (self.n_streams if n is None else n)):
            o = (s * 3 + slot) * self.rgba_stride
            lib().emu_rgba_convert(_ptr(self._slot(s, slot)), g["luma_w"], g["luma_h"], g["width"], g["height"],
                                   _ptr(self.rgba[o:]))

    def read_rgba(self, stream, slot):
        g = self.g
        o = (stream * 3 + slot) * self.rgba_stride
        return self.rgba[o:o + g["width"] * g["height"] * 4].reshape(g["height"], g["width"], 4).copy()

    def close(self):
        pass


class EmuSynth:
    def __init__(self, n_streams=1, fma=0, chunks=1):
        self.n_streams, self.fma, self.chunks = n_streams, fma, chunks
        self.ring = np.zeros((n_streams, 2, 1024), np.float32)
        self.vpos = np.zeros(n_streams, np.int32)
        self.window = (np.array(_window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)

    def synth(self, samples, fmt=0):
        s = np.ascontiguousarray(samples, np.int32)
        n_frames = s.shape[1]
        out = np.zeros((self.n_streams, n_frames, 2304), np.int16 if fmt == desc.AUDIO_S16 else np.float32)
        rc = lib().emu_audio_run(_ptr(s), _ptr(out), _ptr(self.ring), _ptr(self.vpos), _ptr(self.window),
                                 self.n_streams, n_frames, fmt, self.fma, self.chunks)
        assert rc == 0
        return out

    def get_state(self, stream):
        return self.ring[stream].copy(), int(self.vpos[stream])

    def set_state(self, stream, v, vpos):
        if v is not None:
            self.ring[stream] = v
        self.vpos[stream] = vpos


def _window_x2():
    import re
    txt = (ROOT / "mpeg_amd" / "csrc" / "iso11172_synth_window.h").read_text()
    body = txt[txt.index("{") + 1:txt.rindex("}")]
    vals = [int(x) for x in re.findall(r"-?\d+", body)]
    assert len(vals) == 512
    return vals
