"""ctypes binding of include/mpeghip.h (libmpeghip.so) for tests and bench.py.

This is plumbing around the C ABI, not a second implementation: every method is
one ABI call.  There is no CPU path — if the library is not built, or no gfx950
device is present, construction raises MpegHipError."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import desc
from ._build import LIBMPEGHIP

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_RANGE = 0, -1, -2, -3, -4, -5
ABI_VERSION = 3  # include/mpeghip.h: MPEGHIP_ABI_VERSION this binding was written against (checked at load)


class MpegHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libmpeghip error %d: %s" % (code, msg))
        self.code = code


class VideoInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mb_w", C.c_uint32), ("mb_h", C.c_uint32),
                ("luma_w", C.c_uint32), ("luma_h", C.c_uint32), ("chroma_w", C.c_uint32), ("chroma_h", C.c_uint32),
                ("n_streams", C.c_uint32), ("reserved", C.c_uint32),
                ("luma_bytes", C.c_uint64), ("chroma_bytes", C.c_uint64), ("frame_bytes", C.c_uint64),
                ("frame_stride", C.c_uint64), ("rgba_bytes", C.c_uint64)]


# every symbol include/mpeghip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "mpeghip_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "mpeghip_ctx_destroy": (None, [_P]),
    "mpeghip_ctx_sync": (C.c_int, [_P]),
    "mpeghip_device_count": (C.c_int, []),
    "mpeghip_last_error": (C.c_char_p, []),
    "mpeghip_ctx_numa_node": (C.c_int, [_P]),
    "mpeghip_ctx_pci_bus_id": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "mpeghip_abi_version": (C.c_int, []),
    "mpeghip_pinned_alloc": (_P, [_P, C.c_size_t]),
    "mpeghip_pinned_free": (None, [_P, _P]),
    "mpeghip_timer_start": (C.c_int, [_P]),
    "mpeghip_timer_stop_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "mpeghip_video_open": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "mpeghip_video_close": (None, [_P]),
    "mpeghip_video_info_get": (C.c_int, [_P, C.POINTER(VideoInfo)]),
    "mpeghip_video_set_quant": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "mpeghip_video_set_tile_policy": (C.c_int, [_P, C.c_int]),
    "mpeghip_video_submit": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_size_t]),
    "mpeghip_video_stage_begin": (C.c_int, [_P, C.c_uint32, _P, _P, C.POINTER(C.c_void_p)]),
    "mpeghip_video_stage_put": (C.c_int, [_P, C.c_uint32, _P, _P, _P]),
    "mpeghip_video_stage_commit": (C.c_int, [_P]),
    "mpeghip_video_stage_begin_sparse": (C.c_int, [_P, C.c_uint32, _P, _P, C.POINTER(C.c_void_p)]),
    "mpeghip_video_stage_put_sparse": (C.c_int, [_P, C.c_uint32, _P, _P, _P]),
    "mpeghip_video_submit_sparse": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_size_t]),
    "mpeghip_video_stage_begin_device": (C.c_int, [_P, C.c_uint32, _P, _P, C.POINTER(C.c_void_p)]),
    "mpeghip_video_stage_map": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(_P)]),
    "mpeghip_video_stage_put_mapped": (C.c_int, [_P, C.c_uint32, _P]),
    "mpeghip_video_sync": (C.c_int, [_P]),
    "mpeghip_video_verdict": (C.c_int, [_P]),
    "mpeghip_video_refused": (C.c_uint64, [_P, _P, _P, C.c_uint32]),
    "mpeghip_video_batch_upload": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_size_t, C.POINTER(_P)]),
    "mpeghip_video_batch_upload_replicated": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    "mpeghip_video_batch_run": (C.c_int, [_P, _P]),
    "mpeghip_video_batch_free": (None, [_P]),
    "mpeghip_video_batch_alg_bytes": (C.c_uint64, [_P]),
    "mpeghip_video_batch_mbs": (C.c_uint64, [_P]),
    "mpeghip_video_batch_device_bytes": (C.c_uint64, [_P]),
    "mpeghip_video_read_planes": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "mpeghip_video_read_planes_async": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint64)]),
    "mpeghip_video_read_wait": (C.c_int, [_P, C.c_uint64]),
    "mpeghip_video_host_mirror": (C.c_int, [_P, C.c_int]),
    "mpeghip_video_mirror_async": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "mpeghip_video_mirror_counters": (None, [_P, C.POINTER(C.c_uint64 * 2)]),
    "mpeghip_video_write_planes": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P, _P, _P]),
    "mpeghip_video_broadcast_slot": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mpeghip_video_hash_slots": (C.c_int, [_P, C.c_uint32, _P]),
    "mpeghip_video_rgba_convert": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mpeghip_video_read_rgba": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "mpeghip_video_slot_devptr": (_P, [_P, C.c_uint32, C.c_uint32]),
    "mpeghip_video_rgba_devptr": (_P, [_P, C.c_uint32, C.c_uint32]),
    "mpeghip_audio_open": (C.c_int, [_P, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "mpeghip_audio_close": (None, [_P]),
    "mpeghip_audio_synth": (C.c_int, [_P, _P, C.c_uint32, C.c_int, _P]),
    "mpeghip_audio_synth_masked": (C.c_int, [_P, _P, C.c_uint32, C.c_int, _P, _P]),
    "mpeghip_audio_synth_device": (C.c_int, [_P, _P, C.c_uint32, C.c_int, _P]),
    "mpeghip_audio_synth_async": (C.c_int, [_P, _P, C.c_uint32, C.c_int, _P, C.POINTER(C.c_uint64)]),
    "mpeghip_audio_synth_wait": (C.c_int, [_P, C.c_uint64]),
    "mpeghip_audio_undo_last": (C.c_int, [_P]),
    "mpeghip_audio_device_buffers": (C.c_int, [_P, C.c_uint32, C.c_int, C.POINTER(_P), C.POINTER(_P)]),
    "mpeghip_audio_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mpeghip_audio_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mpeghip_audio_get_state": (C.c_int, [_P, C.c_uint32, _P, C.POINTER(C.c_int32)]),
    "mpeghip_audio_set_state": (C.c_int, [_P, C.c_uint32, _P, C.c_int32]),
}

_lib = None


def load_library(path: Path | None = None) -> C.CDLL:
    """dlopen libmpeghip.so (in-tree) and type every declared symbol."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIBMPEGHIP
    if not p.exists():
        raise MpegHipError(ERR_NO_DEVICE, "%s is not built (run __graft_entry__.build()); there is no CPU fallback" % p)
    lib = C.CDLL(str(p))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mpeghip_abi_version() != ABI_VERSION:
        raise MpegHipError(ERR_INVALID, "%s has ABI version %d, this binding is for %d" % (p, lib.mpeghip_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def _check(rc: int):
    if rc != OK:
        raise MpegHipError(rc, load_library().mpeghip_last_error().decode(errors="replace"))


def _ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Context:
    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.mpeghip_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device

    def numa_node(self) -> int:
        """Host NUMA node of the context's GPU (-1: unknown)."""
        return int(self.lib.mpeghip_ctx_numa_node(self.h))

    def pci_bus_id(self) -> str:
        """PCI address of the context's GPU: the identity of the PHYSICAL device (ordinals are per process)."""
        buf = C.create_string_buffer(64)
        _check(self.lib.mpeghip_ctx_pci_bus_id(self.h, buf, 64))
        return buf.value.decode()

    def sync(self):
        _check(self.lib.mpeghip_ctx_sync(self.h))

    def timer_start(self):
        _check(self.lib.mpeghip_timer_start(self.h))

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        _check(self.lib.mpeghip_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def pinned(self, nbytes: int) -> "PinnedBuffer":
        """Pinned host memory of the context (mpeghip_pinned_alloc) as a numpy view: what the asynchronous entries read and write."""
        return PinnedBuffer(self, nbytes)

    def close(self):
        if self.h:
            self.lib.mpeghip_ctx_destroy(self.h)
            self.h = None


class PinnedBuffer:
    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = ctx.lib.mpeghip_pinned_alloc(ctx.h, self.nbytes)
        if not p:
            raise MpegHipError(ERR_INVALID, ctx.lib.mpeghip_last_error().decode(errors="replace"))
        self.ptr = C.c_void_p(p)
        self.u8 = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(self.nbytes,))

    def view(self, dtype):
        return self.u8.view(dtype)

    def free(self):
        if self.ptr:
            self.u8 = None
            self.ctx.lib.mpeghip_pinned_free(self.ctx.h, self.ptr)
            self.ptr = None


class Batch:
    def __init__(self, video: "VideoStore", h):
        self.video, self.h = video, h
        self.alg_bytes = video.lib.mpeghip_video_batch_alg_bytes(h)
        self.n_mbs = video.lib.mpeghip_video_batch_mbs(h)
        self.device_bytes = video.lib.mpeghip_video_batch_device_bytes(h)

    def run(self):
        _check(self.video.lib.mpeghip_video_batch_run(self.video.h, self.h))

    def free(self):
        if self.h:
            self.video.lib.mpeghip_video_batch_free(self.h)
            self.h = None


class VideoStore:
    """Frame store + reconstruction for n_streams independent streams of one size."""

    def __init__(self, ctx: Context, width: int, height: int, n_streams: int = 1):
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        _check(self.lib.mpeghip_video_open(ctx.h, width, height, n_streams, C.byref(h)))
        self.h = h
        self.info = VideoInfo()
        _check(self.lib.mpeghip_video_info_get(h, C.byref(self.info)))
        self.n_streams = n_streams

    def set_quant(self, stream, intra, non_intra):
        i = np.ascontiguousarray(intra, dtype=np.uint8)
        n = np.ascontiguousarray(non_intra, dtype=np.uint8)
        _check(self.lib.mpeghip_video_set_quant(self.h, stream, _ptr(i), _ptr(n)))

    def set_tile_policy(self, policy: int):
        """0 = the library picks the kernel instance per batch, 1 = int16 tile (8 waves per SIMD), 2 = int32 tile."""
        _check(self.lib.mpeghip_video_set_tile_policy(self.h, policy))

    @staticmethod
    def _args(pics, mbs, coefs):
        pics = np.ascontiguousarray(pics, dtype=desc.PIC_DTYPE)
        mbs = np.ascontiguousarray(mbs, dtype=desc.MB_DTYPE)
        coefs = np.ascontiguousarray(coefs).view(np.uint8).reshape(-1)
        return pics, mbs, coefs

    def submit(self, pics, mbs, coefs):
        pics, mbs, coefs = self._args(pics, mbs, coefs)
        _check(self.lib.mpeghip_video_submit(self.h, _ptr(pics), len(pics), _ptr(mbs), len(mbs), _ptr(coefs), coefs.nbytes))

    def submit_staged(self, pictures, threads: int = 1):
        """One submit assembled picture by picture (mpeghip_video_stage_*): pictures = [(pic, mbs, coefs)], each
        with coef_off relative to its own coefs; the puts run on `threads` host threads."""
        parts = []
        for pic, mbs, coefs in pictures:
            p, m, c = self._args(np.asarray(pic).reshape(1), mbs, coefs)
            parts.append((p, m, c))
        n_mbs = np.array([len(m) for _, m, _ in parts], np.uint32)
        nbytes = np.array([c.nbytes for _, _, c in parts], np.uint64)  # size_t
        st = C.c_void_p()
        _check(self.lib.mpeghip_video_stage_begin(self.h, len(parts), _ptr(n_mbs), _ptr(nbytes), C.byref(st)))
        rcs = [0] * len(parts)

        def put(i):
            p, m, c = parts[i]
            rcs[i] = self.lib.mpeghip_video_stage_put(st, i, _ptr(p), _ptr(m), _ptr(c))

        if threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(put, range(len(parts))))
        else:
            for i in range(len(parts)):
                put(i)
        _check(self.lib.mpeghip_video_stage_commit(st))  # reports the first failed put, launches nothing then
        return rcs

    def submit_sparse(self, pic, mbs, words):
        """One picture in the sparse hand-over form (desc.to_sparse)."""
        p = np.ascontiguousarray(np.asarray(pic).reshape(1), dtype=desc.PIC_DTYPE)
        m = np.ascontiguousarray(mbs, dtype=desc.MB_DTYPE)
        w = np.ascontiguousarray(words, dtype=np.uint32)
        _check(self.lib.mpeghip_video_submit_sparse(self.h, _ptr(p), _ptr(m), len(m), _ptr(w), len(w)))

    def submit_staged_sparse(self, pictures, threads: int = 1):
        """submit_staged with every picture in the sparse form: pictures = [(pic, mbs, words)]."""
        parts = [(np.ascontiguousarray(np.asarray(p).reshape(1), dtype=desc.PIC_DTYPE), np.ascontiguousarray(m, dtype=desc.MB_DTYPE),
                  np.ascontiguousarray(w, dtype=np.uint32)) for p, m, w in pictures]
        n_mbs = np.array([len(m) for _, m, _ in parts], np.uint32)
        n_words = np.array([len(w) for _, _, w in parts], np.uint64)  # size_t
        st = C.c_void_p()
        _check(self.lib.mpeghip_video_stage_begin_sparse(self.h, len(parts), _ptr(n_mbs), _ptr(n_words), C.byref(st)))
        rcs = [0] * len(parts)

        def put(i):
            p, m, w = parts[i]
            rcs[i] = self.lib.mpeghip_video_stage_put_sparse(st, i, _ptr(p), _ptr(m), _ptr(w))

        if threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(put, range(len(parts))))
        else:
            for i in range(len(parts)):
                put(i)
        _check(self.lib.mpeghip_video_stage_commit(st))
        return rcs

    def submit_staged_device(self, pictures, threads: int = 1, mapped: bool = False, sync: bool = True):
        """A DEVICE-PACKED stage (mpeghip_video_stage_begin_device): pictures = [(pic, mbs, words)] in the sparse form; the host
        only copies them (mapped: writes them straight into the staging buffer through mpeghip_video_stage_map), the device
        validates and packs.  sync: wait and raise the commit's deferred error, if any (else the caller does: self.sync())."""
        parts = [(np.ascontiguousarray(np.asarray(p).reshape(1), dtype=desc.PIC_DTYPE), np.ascontiguousarray(m, dtype=desc.MB_DTYPE),
                  np.ascontiguousarray(w, dtype=np.uint32)) for p, m, w in pictures]
        n_mbs = np.array([len(m) for _, m, _ in parts], np.uint32)
        n_words = np.array([len(w) for _, _, w in parts], np.uint64)  # size_t
        st = C.c_void_p()
        _check(self.lib.mpeghip_video_stage_begin_device(self.h, len(parts), _ptr(n_mbs), _ptr(n_words), C.byref(st)))
        rcs = [0] * len(parts)

        def put(i):
            p, m, w = parts[i]
            if mapped:
                pm, pw = C.c_void_p(), C.c_void_p()
                rcs[i] = self.lib.mpeghip_video_stage_map(st, i, C.byref(pm), C.byref(pw))
                if rcs[i] == OK:
                    if len(m):
                        C.memmove(pm, m.ctypes.data, m.nbytes)
                    if len(w):
                        C.memmove(pw, w.ctypes.data, w.nbytes)
                    rcs[i] = self.lib.mpeghip_video_stage_put_mapped(st, i, _ptr(p))
            else:
                rcs[i] = self.lib.mpeghip_video_stage_put_sparse(st, i, _ptr(p), _ptr(m), _ptr(w))

        if threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(put, range(len(parts))))
        else:
            for i in range(len(parts)):
                put(i)
        _check(self.lib.mpeghip_video_stage_commit(st))
        if sync:
            self.sync()
        return rcs

    def sync(self):
        """mpeghip_video_sync: wait for the handle's queued work; raises the deferred error of a device-packed commit."""
        _check(self.lib.mpeghip_video_sync(self.h))

    def verdict(self):
        """mpeghip_video_verdict: wait for the VALIDATION of the device-packed commits queued so far (not their reconstruction);
        raises their deferred error."""
        _check(self.lib.mpeghip_video_verdict(self.h))

    def refused(self, cap: int = 1024):
        """-> (how many pictures the last reported verdict refused, [(picture index in its commit, stream), ...])"""
        pics, streams = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        n = int(self.lib.mpeghip_video_refused(self.h, _ptr(pics), _ptr(streams), cap))
        k = min(n, cap)
        return n, list(zip(pics[:k].tolist(), streams[:k].tolist()))

    def upload(self, pics, mbs, coefs, replicate: int = 1) -> Batch:
        pics, mbs, coefs = self._args(pics, mbs, coefs)
        h = C.c_void_p()
        _check(self.lib.mpeghip_video_batch_upload_replicated(self.h, _ptr(pics), len(pics), _ptr(mbs), len(mbs),
                                                              _ptr(coefs), coefs.nbytes, replicate, C.byref(h)))
        return Batch(self, h)

    def read_planes(self, stream: int, slot: int):
        y = np.empty(self.info.luma_bytes, np.uint8)
        cb = np.empty(self.info.chroma_bytes, np.uint8)
        cr = np.empty(self.info.chroma_bytes, np.uint8)
        _check(self.lib.mpeghip_video_read_planes(self.h, stream, slot, _ptr(y), _ptr(cb), _ptr(cr)))
        return y, cb, cr

    def read_planes_async(self, stream: int, slot: int, pinned: "PinnedBuffer") -> int:
        """Queue the read-back of (stream, slot) into `pinned` (Context.pinned: luma | Cb | Cr, linear) -> ticket."""
        assert pinned.nbytes >= self.info.luma_bytes + 2 * self.info.chroma_bytes
        t = C.c_uint64()
        _check(self.lib.mpeghip_video_read_planes_async(self.h, stream, slot, pinned.ptr, C.byref(t)))
        return t.value

    def read_wait(self, ticket: int):
        _check(self.lib.mpeghip_video_read_wait(self.h, ticket))

    def host_mirror(self, on: bool = True):
        """mpeghip_video_host_mirror: every (stream, slot)'s planes once more, linear, in pinned host memory, written by the
        reconstruction launches themselves (small submits: the library's four-waves-per-chunk kernel)."""
        _check(self.lib.mpeghip_video_host_mirror(self.h, 1 if on else 0))

    def mirror_counters(self):
        """-> (mirror_async calls, those that had to untile the slot first)"""
        out = (C.c_uint64 * 2)()
        self.lib.mpeghip_video_mirror_counters(self.h, C.byref(out))
        return out[0], out[1]

    def mirror_async(self, stream: int, slot: int):
        """-> (numpy view of the slot's linear copy in pinned host memory: luma | Cb | Cr, ticket); the view holds the slot as it
        is after everything submitted so far once read_wait(ticket) has returned."""
        p, t = _P(), C.c_uint64()
        _check(self.lib.mpeghip_video_mirror_async(self.h, stream, slot, C.byref(p), C.byref(t)))
        n = self.info.luma_bytes + 2 * self.info.chroma_bytes
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)), t.value

    def split_planes(self, flat):
        L, Cb = self.info.luma_bytes, self.info.chroma_bytes
        return flat[:L], flat[L:L + Cb], flat[L + Cb:L + 2 * Cb]

    def write_planes(self, stream: int, slot: int, y, cb, cr, pad=None):
        y, cb, cr = (np.ascontiguousarray(a, np.uint8) for a in (y, cb, cr))
        pad = None if pad is None else np.ascontiguousarray(pad, np.uint8)
        _check(self.lib.mpeghip_video_write_planes(self.h, stream, slot, _ptr(y), _ptr(cb), _ptr(cr), _ptr(pad)))

    def broadcast_slot(self, src: int, slot: int, dst0: int, n: int):
        _check(self.lib.mpeghip_video_broadcast_slot(self.h, src, slot, dst0, n))

    def hash_slots(self, slot: int) -> np.ndarray:
        out = np.empty(self.n_streams, np.uint64)
        _check(self.lib.mpeghip_video_hash_slots(self.h, slot, _ptr(out)))
        return out

    def rgba_convert(self, slot: int, stream0: int = 0, n: int | None = None):
        _check(self.lib.mpeghip_video_rgba_convert(self.h, slot, stream0, self.n_streams if n is None else n))

    def read_rgba(self, stream: int, slot: int) -> np.ndarray:
        out = np.empty(self.info.rgba_bytes, np.uint8)
        _check(self.lib.mpeghip_video_read_rgba(self.h, stream, slot, _ptr(out)))
        return out.reshape(self.info.height, self.info.width, 4)

    def close(self):
        if self.h:
            self.lib.mpeghip_video_close(self.h)
            self.h = None


class AudioSynth:
    """MP2 sub-band synthesis for n_streams independent streams."""

    def __init__(self, ctx: Context, n_streams: int = 1, fma: int = desc.AUDIO_FMA_NONE):
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        _check(self.lib.mpeghip_audio_open(ctx.h, n_streams, fma, C.byref(h)))
        self.h, self.n_streams = h, n_streams

    @staticmethod
    def out_dtype(fmt):
        return np.int16 if fmt == desc.AUDIO_S16 else np.float32

    def synth(self, samples: np.ndarray, fmt: int = desc.AUDIO_F32N) -> np.ndarray:
        """samples int32 [n_streams, n_frames, 2, 36, 32] -> [n_streams, n_frames, 2304]."""
        s = np.ascontiguousarray(samples, dtype=np.int32)
        assert s.shape[0] == self.n_streams and s.shape[2:] == (2, 36, 32)
        n_frames = s.shape[1]
        out = np.empty((self.n_streams, n_frames, 2304), self.out_dtype(fmt))
        _check(self.lib.mpeghip_audio_synth(self.h, _ptr(s), n_frames, fmt, _ptr(out)))
        return out

    def synth_async(self, pinned_in: "PinnedBuffer", n_frames: int, fmt: int, pinned_out: "PinnedBuffer") -> int:
        """Queue the synthesis of the samples in `pinned_in` (int32 [n_streams, n_frames, 2, 36, 32]) into `pinned_out` -> ticket."""
        t = C.c_uint64()
        _check(self.lib.mpeghip_audio_synth_async(self.h, pinned_in.ptr, n_frames, fmt, pinned_out.ptr, C.byref(t)))
        return t.value

    def synth_wait(self, ticket: int):
        _check(self.lib.mpeghip_audio_synth_wait(self.h, ticket))

    def undo_last(self):
        _check(self.lib.mpeghip_audio_undo_last(self.h))

    def synth_masked(self, samples: np.ndarray, active, fmt: int = desc.AUDIO_F32N, out=None) -> np.ndarray:
        """Like synth() for the streams with active[i] != 0; the others keep their state and their rows of `out`."""
        s = np.ascontiguousarray(samples, dtype=np.int32)
        assert s.shape[0] == self.n_streams and s.shape[2:] == (2, 36, 32)
        n_frames = s.shape[1]
        if out is None:
            out = np.zeros((self.n_streams, n_frames, 2304), self.out_dtype(fmt))
        mask = np.ascontiguousarray(active, dtype=np.uint8)
        assert mask.shape == (self.n_streams,)
        tmp = np.empty_like(out)
        _check(self.lib.mpeghip_audio_synth_masked(self.h, _ptr(s), n_frames, fmt, _ptr(tmp), _ptr(mask)))
        out[mask != 0] = tmp[mask != 0]
        return out

    def device_buffers(self, n_frames: int, fmt: int):
        ds, do = C.c_void_p(), C.c_void_p()
        _check(self.lib.mpeghip_audio_device_buffers(self.h, n_frames, fmt, C.byref(ds), C.byref(do)))
        return ds, do

    def upload(self, d_dst, samples: np.ndarray):
        s = np.ascontiguousarray(samples, dtype=np.int32)
        _check(self.lib.mpeghip_audio_upload(self.h, d_dst, _ptr(s), s.size))

    def download(self, d_src, n_elems: int, fmt: int) -> np.ndarray:
        out = np.empty(n_elems, self.out_dtype(fmt))
        _check(self.lib.mpeghip_audio_download(self.h, _ptr(out), d_src, out.nbytes))
        return out

    def synth_device(self, d_samples, n_frames: int, fmt: int, d_out):
        _check(self.lib.mpeghip_audio_synth_device(self.h, d_samples, n_frames, fmt, d_out))

    def get_state(self, stream: int):
        v = np.empty((2, 1024), np.float32)
        vpos = C.c_int32()
        _check(self.lib.mpeghip_audio_get_state(self.h, stream, _ptr(v), C.byref(vpos)))
        return v, vpos.value

    def set_state(self, stream: int, v, vpos: int):
        v = None if v is None else np.ascontiguousarray(v, np.float32)
        _check(self.lib.mpeghip_audio_set_state(self.h, stream, _ptr(v), vpos))

    def close(self):
        if self.h:
            self.lib.mpeghip_audio_close(self.h)
            self.h = None
