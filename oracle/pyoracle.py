"""ctypes wrapper of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  The product (mpeg_amd/) never does."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"
FNV_OFFSET = 0xcbf29ce484222325

PIC_DTYPE = np.dtype([("stream", "<u4"), ("cur", "u1"), ("fwd", "u1"), ("bwd", "u1"), ("flags", "u1"),
                      ("mb_first", "<u4"), ("mb_count", "<u4")])


class Frame(C.Structure):
    _fields_ = [("base", C.c_void_p), ("total", C.c_size_t), ("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p),
                ("luma_size", C.c_size_t), ("chroma_size", C.c_size_t), ("luma_w", C.c_int), ("luma_h", C.c_int),
                ("chroma_w", C.c_int), ("chroma_h", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("time", C.c_double)]


class VideoStats(C.Structure):
    _fields_ = [("pictures", C.c_int * 4), ("frames_returned", C.c_int), ("invalid_blocks", C.c_int),
                ("coded_blocks", C.c_int), ("dc_only_blocks", C.c_int), ("sparse_idct", C.c_int), ("full_idct", C.c_int),
                ("coded_mbs", C.c_int), ("intra_mbs", C.c_int), ("skipped_mbs", C.c_int), ("bidir_mbs", C.c_int),
                ("copy_mb_calls", C.c_int), ("copy_mode", C.c_int * 4), ("overreads", C.c_int), ("range_errors", C.c_int),
                ("max_idct_in", C.c_int64), ("max_idct_out", C.c_int64), ("max_idct_mid", C.c_int64), ("max_abs_mv", C.c_int)]


class Synth(C.Structure):
    _fields_ = [("v", (C.c_float * 1024) * 2), ("vpos", C.c_int32)]


def build(force: bool = False) -> Path:
    srcs = [HERE / n for n in ("mpeg_oracle.c", "oracle_desc.c", "mpeg_oracle.h", "oracle_desc.h",
                               "iso11172_vlc_codes.h", "iso11172_synth_window.h")] + [HERE.parent / "include" / "mpeghip.h"]
    if not force and LIB.exists() and all(s.stat().st_mtime <= LIB.stat().st_mtime for s in srcs):
        return LIB
    r = subprocess.run(["make", "-C", str(HERE), "-B", "liboracle.so"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB))
        P = C.c_void_p
        sig = {
            "orc_fnv1a64": (C.c_uint64, [C.c_uint64, P, C.c_size_t]),
            "orc_idct": (C.c_int64, [P, C.c_int]),
            "orc_frame_alloc": (C.c_int, [C.POINTER(Frame), C.c_int, C.c_int]),
            "orc_frame_free": (None, [C.POINTER(Frame)]),
            "orc_copy_macroblock": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Frame), C.POINTER(Frame)]),
            "orc_copy_macroblock_ref": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Frame), C.POINTER(Frame)]),
            "orc_test_frame_fill": (None, [C.POINTER(Frame), C.c_int]),
            "orc_idct36": (None, [P, C.c_int, P, C.c_int]),
            "orc_synth_window": (None, [P, P, P, C.c_int, C.c_int]),
            "orc_synth_window_ref": (None, [P, P, P, C.c_int, C.c_int]),
            "orc_window_table": (None, [P]),
            "orc_ycbcr_to_rgba": (None, [C.POINTER(Frame), P]),
            "orc_video_open": (P, [C.c_char_p, C.c_size_t]),
            "orc_video_close": (None, [P]),
            "orc_video_has_header": (C.c_int, [P]),
            "orc_video_width": (C.c_int, [P]),
            "orc_video_height": (C.c_int, [P]),
            "orc_video_framerate": (C.c_double, [P]),
            "orc_video_set_no_delay": (None, [P, C.c_int]),
            "orc_video_decode": (C.POINTER(Frame), [P]),
            "orc_video_rewind": (None, [P]), "orc_video_time": (C.c_double, [P]), "orc_video_has_ended": (C.c_int, [P]),
            "orc_audio_rewind": (None, [P]), "orc_audio_time": (C.c_double, [P]), "orc_audio_has_ended": (C.c_int, [P]),
            "orc_video_get_stats": (C.POINTER(VideoStats), [P]),
            "orc_audio_open": (P, [C.c_char_p, C.c_size_t, C.c_int]),
            "orc_audio_close": (None, [P]),
            "orc_audio_samplerate": (C.c_int, [P]),
            "orc_audio_channels": (C.c_int, [P]),
            "orc_audio_decode": (C.POINTER(C.c_float), [P, P]),
            "orc_ps_extract": (P, [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
            "orc_store_open": (P, [C.c_int, C.c_int, C.c_uint32]),
            "orc_store_close": (None, [P]),
            "orc_store_frame": (C.POINTER(Frame), [P, C.c_uint32, C.c_uint32]),
            "orc_store_set_quant": (None, [P, C.c_uint32, P, P]),
            "orc_store_submit": (C.c_int, [P, P, C.c_uint32, P, C.c_uint32, P, C.c_size_t, C.c_int]),
            "orc_synth_frames": (None, [C.POINTER(Synth), P, C.c_uint32, C.c_int, C.c_int, P]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fnv1a64(data, h: int = FNV_OFFSET) -> int:
    a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return lib().orc_fnv1a64(h, _ptr(a), a.nbytes)


def frame_planes(f: Frame):
    """Copies of (Y, Cb, Cr) of an oracle frame."""
    y = np.ctypeslib.as_array(C.cast(f.y, C.POINTER(C.c_uint8)), shape=(f.luma_size,)).copy()
    cb = np.ctypeslib.as_array(C.cast(f.cb, C.POINTER(C.c_uint8)), shape=(f.chroma_size,)).copy()
    cr = np.ctypeslib.as_array(C.cast(f.cr, C.POINTER(C.c_uint8)), shape=(f.chroma_size,)).copy()
    return y, cb, cr


class VideoDecoder:
    """NewVideo + Video.Decode of the reference, restated on the CPU."""

    def __init__(self, data: bytes):
        self._data = data  # keep alive: the oracle reads it in place
        self.h = lib().orc_video_open(data, len(data))

    width = property(lambda s: lib().orc_video_width(s.h))
    height = property(lambda s: lib().orc_video_height(s.h))
    framerate = property(lambda s: lib().orc_video_framerate(s.h))

    def decode(self):
        f = lib().orc_video_decode(self.h)
        return f.contents if f else None

    def stats(self) -> VideoStats:
        return lib().orc_video_get_stats(self.h).contents

    def rewind(self):
        lib().orc_video_rewind(self.h)

    time = property(lambda s: lib().orc_video_time(s.h))
    has_ended = property(lambda s: bool(lib().orc_video_has_ended(s.h)))

    def close(self):
        if self.h:
            lib().orc_video_close(self.h)
            self.h = None


class AudioDecoder:
    def __init__(self, data: bytes, fma: int = 0):
        self._data = data
        self.h = lib().orc_audio_open(data, len(data), fma)

    samplerate = property(lambda s: lib().orc_audio_samplerate(s.h))
    channels = property(lambda s: lib().orc_audio_channels(s.h))

    def decode(self, want_samples: bool = False):
        """Returns interleaved float32[2304] (and the int32 [2,36,32] sub-band samples) or None."""
        samples = np.zeros((2, 36, 32), np.int32) if want_samples else None
        p = lib().orc_audio_decode(self.h, _ptr(samples))
        if not p:
            return None
        out = np.ctypeslib.as_array(p, shape=(2304,)).copy()
        return (out, samples) if want_samples else out

    def rewind(self):
        lib().orc_audio_rewind(self.h)

    time = property(lambda s: lib().orc_audio_time(s.h))
    has_ended = property(lambda s: bool(lib().orc_audio_has_ended(s.h)))

    def close(self):
        if self.h:
            lib().orc_audio_close(self.h)
            self.h = None


def ps_extract(data: bytes, packet_type: int):
    n, k = C.c_size_t(), C.c_int()
    p = lib().orc_ps_extract(data, len(data), packet_type, C.byref(n), C.byref(k))
    out = C.string_at(p, n.value)
    lib().free(p)
    return out, k.value


class OracleStore:
    """Same surface as mpeg_amd.abi.VideoStore, computed by the reference restatement."""

    def __init__(self, width: int, height: int, n_streams: int = 1, threads: int = 1):
        self.h = lib().orc_store_open(width, height, n_streams)
        self.n_streams, self.threads = n_streams, threads
        self.width, self.height = width, height

    def frame(self, stream: int, slot: int) -> Frame:
        return lib().orc_store_frame(self.h, stream, slot).contents

    def set_quant(self, stream, intra, non_intra):
        i = np.ascontiguousarray(intra, np.uint8)
        n = np.ascontiguousarray(non_intra, np.uint8)
        lib().orc_store_set_quant(self.h, stream, _ptr(i), _ptr(n))

    def submit(self, pics, mbs, coefs):
        pics = np.ascontiguousarray(pics)
        mbs = np.ascontiguousarray(mbs)
        coefs = np.ascontiguousarray(coefs).view(np.uint8).reshape(-1)
        rc = lib().orc_store_submit(self.h, _ptr(pics), len(pics), _ptr(mbs), len(mbs), _ptr(coefs), coefs.nbytes, self.threads)
        if rc != 0:
            raise RuntimeError("oracle: motion vector outside the frame buffer (the reference would panic)")

    def read_planes(self, stream: int, slot: int):
        return frame_planes(self.frame(stream, slot))

    def write_planes(self, stream: int, slot: int, y, cb, cr, pad=None):
        f = self.frame(stream, slot)
        C.memmove(f.y, np.ascontiguousarray(y, np.uint8).ctypes.data, f.luma_size)
        C.memmove(f.cb, np.ascontiguousarray(cb, np.uint8).ctypes.data, f.chroma_size)
        C.memmove(f.cr, np.ascontiguousarray(cr, np.uint8).ctypes.data, f.chroma_size)
        if pad is not None:
            C.memmove(f.cr + f.chroma_size, np.ascontiguousarray(pad, np.uint8).ctypes.data, f.luma_w * 16)

    def read_rgba(self, stream: int, slot: int) -> np.ndarray:
        f = self.frame(stream, slot)
        out = np.empty((self.height, self.width, 4), np.uint8)
        lib().orc_ycbcr_to_rgba(C.byref(f), _ptr(out))
        return out

    def close(self):
        if self.h:
            lib().orc_store_close(self.h)
            self.h = None


class OracleSynth:
    """audio.go:378-422 for n_streams independent streams (V ring + vPos each)."""

    def __init__(self, n_streams: int = 1, fma: int = 0):
        self.states = [Synth() for _ in range(n_streams)]
        self.fma = fma

    def synth(self, samples: np.ndarray, fmt: int = 0) -> np.ndarray:
        s = np.ascontiguousarray(samples, np.int32)
        n_streams, n_frames = s.shape[0], s.shape[1]
        out = np.empty((n_streams, n_frames, 2304), np.int16 if fmt == 3 else np.float32)
        for i in range(n_streams):
            lib().orc_synth_frames(C.byref(self.states[i]), _ptr(s[i]), n_frames, fmt, self.fma, _ptr(out[i]))
        return out

    def set_state(self, stream: int, v, vpos: int):
        st = self.states[stream]
        if v is not None:
            np.ctypeslib.as_array(st.v).reshape(2, 1024)[:] = np.asarray(v, np.float32).reshape(2, 1024)
        st.vpos = vpos

    def get_state(self, stream: int):
        st = self.states[stream]
        return np.ctypeslib.as_array(st.v).reshape(2, 1024).copy(), int(st.vpos)
