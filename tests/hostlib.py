"""ctypes face of the product's host library (mpeg_amd/libmpeghost.so: bitstream parse -> descriptors ->
libmpeghip) and of tests/host_emu (TEST-ONLY backend built on the lane emulator, for CPU-side parser checks)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
HOST_EMU = ROOT / "tests" / "host_emu" / "libhost_emu.so"


class HostFrame(C.Structure):
    _fields_ = [("time", C.c_double), ("width", C.c_int), ("height", C.c_int), ("luma_w", C.c_int), ("luma_h", C.c_int),
                ("chroma_w", C.c_int), ("chroma_h", C.c_int), ("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p),
                ("luma_bytes", C.c_size_t), ("chroma_bytes", C.c_size_t)]


_host = None
_emu = None


def host():
    global _host
    if _host is None:
        from mpeg_amd import _build
        _build.build_libmpeghost()
        L = C.CDLL(str(_build.LIBMPEGHOST))
        P = C.c_void_p
        sig = {
            "mpeghost_last_error": (C.c_char_p, []),
            "mpeghost_device_create": (P, [C.c_int]), "mpeghost_device_destroy": (None, [P]),
            "mpeghost_video_open": (P, [P, C.c_char_p, C.c_size_t]),
            "mpeghost_video_open_backend": (P, [P, C.c_char_p, C.c_size_t]),
            "mpeghost_video_close": (None, [P]), "mpeghost_video_width": (C.c_int, [P]), "mpeghost_video_height": (C.c_int, [P]),
            "mpeghost_video_framerate": (C.c_double, [P]), "mpeghost_video_set_no_delay": (None, [P, C.c_int]),
            "mpeghost_video_set_sparse": (None, [P, C.c_int]), "mpeghost_set_default_sparse": (None, [C.c_int]),
            "mpeghost_debug_vlc_self_check": (C.c_uint64, []),
            "mpeghost_debug_vlc_decode": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "mpeghost_video_decode": (C.c_int, [P, C.POINTER(HostFrame)]), "mpeghost_video_rgba": (P, [P]),
            "mpeghost_video_stats": (None, [P, C.POINTER(C.c_uint64 * 8)]),
            "mpeghost_video_phase_seconds": (None, [P, C.POINTER(C.c_double * 3)]),
            "mpeghost_video_time": (C.c_double, [P]), "mpeghost_video_has_ended": (C.c_int, [P]), "mpeghost_video_rewind": (None, [P]),
            "mpeghost_video_set_lookahead": (None, [P, C.c_int]), "mpeghost_video_set_host_mirror": (None, [P, C.c_int]),
            "mpeghost_video_set_device_pack_from": (None, [P, C.c_uint32]),
            "mpeghost_audio_time": (C.c_double, [P]), "mpeghost_audio_has_ended": (C.c_int, [P]), "mpeghost_audio_rewind": (None, [P]),
            "mpeghost_audio_set_lookahead": (None, [P, C.c_int]),
            "mpeghost_audio_open": (P, [P, C.c_char_p, C.c_size_t, C.c_int, C.c_int]),
            "mpeghost_audio_open_backend": (P, [P, C.c_char_p, C.c_size_t, C.c_int]),
            "mpeghost_audio_close": (None, [P]), "mpeghost_audio_samplerate": (C.c_int, [P]), "mpeghost_audio_channels": (C.c_int, [P]),
            "mpeghost_audio_decode": (P, [P, C.POINTER(C.c_double)]),
            "mpeghost_mpeg_open": (P, [P, C.c_char_p, C.c_size_t]), "mpeghost_mpeg_close": (None, [P]),
            "mpeghost_mpeg_info": (None, [P, C.POINTER(C.c_int * 6)]), "mpeghost_mpeg_framerate": (C.c_double, [P]),
            "mpeghost_mpeg_set_enabled": (None, [P, C.c_int, C.c_int]),
            "mpeghost_mpeg_get_enabled": (None, [P, C.POINTER(C.c_int * 2)]),
            "mpeghost_mpeg_set_audio_stream": (None, [P, C.c_int]),
            "mpeghost_mpeg_set_loop": (None, [P, C.c_int]), "mpeghost_mpeg_loop": (C.c_int, [P]),
            "mpeghost_mpeg_rewind": (None, [P]),
            "mpeghost_mpeg_decode_video": (C.c_int, [P, C.POINTER(HostFrame)]),
            "mpeghost_mpeg_decode_audio": (P, [P, C.POINTER(C.c_double)]), "mpeghost_mpeg_has_ended": (C.c_int, [P]),
            "mpeghost_mpeg_take_done": (C.c_int, [P]), "mpeghost_mpeg_audio_format": (C.c_int, [P]),
            "mpeghost_mpeg_set_audio_format": (None, [P, C.c_int]), "mpeghost_mpeg_audio_lead_time": (C.c_double, [P]),
            "mpeghost_mpeg_set_audio_lead_time": (None, [P, C.c_double]),
            "mpeghost_mpeg_open_backends": (P, [P, P, C.c_char_p, C.c_size_t]),
            "mpeghost_mpeg_probe": (C.c_int, [P, C.c_size_t]), "mpeghost_mpeg_has_headers": (C.c_int, [P]),
            "mpeghost_mpeg_duration": (C.c_double, [P]), "mpeghost_mpeg_time": (C.c_double, [P]),
            "mpeghost_mpeg_audio_time": (C.c_double, [P]), "mpeghost_mpeg_video_time": (C.c_double, [P]),
            "mpeghost_mpeg_count_callbacks": (None, [P, C.c_int, C.c_int]),
            "mpeghost_mpeg_callback_counts": (None, [P, C.POINTER(C.c_int * 2)]),
            "mpeghost_mpeg_decode": (None, [P, C.c_double]),
            "mpeghost_mpeg_seek": (C.c_int, [P, C.c_double, C.c_int]),
            "mpeghost_mpeg_seek_frame": (C.c_int, [P, C.c_double, C.c_int, C.POINTER(HostFrame)]),
            "mpeghost_batch_open": (P, [P, C.c_uint32]), "mpeghost_batch_open_store": (P, [P, C.c_uint32]),
            "mpeghost_batch_close": (None, [P]), "mpeghost_batch_add_stream": (C.c_int, [P, C.c_char_p, C.c_size_t]),
            "mpeghost_batch_decode_all": (C.c_int, [P, C.c_int]), "mpeghost_batch_set_threads": (None, [P, C.c_uint32]),
            "mpeghost_batch_frame": (C.c_int, [P, C.c_uint32, C.POINTER(HostFrame)]),
            "mpeghost_batch_counters": (None, [P, C.POINTER(C.c_uint64 * 2)]),
            "mpeghost_batch_phase_seconds": (None, [P, C.POINTER(C.c_double * 4)]),
            "mpeghost_batch_set_device_pack": (None, [P, C.c_int]), "mpeghost_batch_sync": (C.c_int, [P]),
            "mpeghost_batch_numa_pins": (None, [P, C.POINTER(C.c_uint32 * 2)]),
            "mpeghost_batch_threads": (C.c_uint32, [P]), "mpeghost_effective_cores": (C.c_double, []),
            "mpeghost_cgroup_quota_cores": (C.c_double, [C.c_char_p, C.c_char_p]),
            "mpeghost_batch_refused_streams": (C.c_uint32, [P, C.POINTER(C.c_uint32), C.c_uint32]), "mpeghost_batch_device_pack": (C.c_int, [P]),
            "mpeghost_batch_debug_damage_next_picture": (None, [P, C.c_uint32]), "mpeghost_sharded_threads": (C.c_uint32, [P]),
            "mpeghost_sharded_set_device_pack": (None, [P, C.c_int]), "mpeghost_sharded_sync": (C.c_int, [P]),
            "mpeghost_sharded_open": (P, [C.POINTER(P), C.c_uint32, C.c_uint32]),
            "mpeghost_sharded_open_stores": (P, [C.POINTER(P), C.c_uint32, C.c_uint32]),
            "mpeghost_sharded_close": (None, [P]), "mpeghost_sharded_add_stream": (C.c_int, [P, C.c_char_p, C.c_size_t]),
            "mpeghost_sharded_decode_all": (C.c_int, [P, C.c_int]), "mpeghost_sharded_set_threads": (None, [P, C.c_uint]),
            "mpeghost_sharded_frame": (C.c_int, [P, C.c_uint32, C.POINTER(HostFrame)]),
            "mpeghost_sharded_device_of": (C.c_uint32, [P, C.c_uint32]),
            "mpeghost_sharded_counters": (None, [P, C.c_uint32, C.POINTER(C.c_uint64 * 2)]),
            "mpeghost_audio_batch_open": (P, [P, C.c_uint32, C.c_int, C.c_int]),
            "mpeghost_audio_batch_open_store": (P, [P, C.c_uint32, C.c_int, C.c_int]),
            "mpeghost_audio_batch_close": (None, [P]), "mpeghost_audio_batch_add_stream": (C.c_int, [P, C.c_char_p, C.c_size_t]),
            "mpeghost_audio_batch_decode_all": (C.c_int, [P]),
            "mpeghost_audio_batch_decode_stream": (C.c_int, [P, C.c_uint32]),
            "mpeghost_audio_batch_samples": (P, [P, C.c_uint32, C.POINTER(C.c_double), C.POINTER(P)]),
            "mpeghost_audio_batch_device_calls": (C.c_uint64, [P]),
            "mpeghost_audio_batch_set_threads": (None, [P, C.c_uint32]),
            "mpeghost_demux_open": (P, [C.c_char_p, C.c_size_t]), "mpeghost_demux_close": (None, [P]),
            "mpeghost_demux_start_time": (C.c_double, [P, C.c_int]), "mpeghost_demux_duration": (C.c_double, [P, C.c_int]),
            "mpeghost_demux_probe": (C.c_int, [P, C.c_size_t]), "mpeghost_demux_streams": (None, [P, C.POINTER(C.c_int * 2)]),
            "mpeghost_demux_rewind": (None, [P]),
            "mpeghost_demux_decode": (C.c_int, [P, C.POINTER(C.c_double), C.POINTER(C.c_size_t), C.POINTER(P)]),
            "mpeghost_demux_seek": (C.c_int, [P, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_size_t), C.POINTER(P)]),
        }
        for n, (r, a) in sig.items():
            f = getattr(L, n)
            f.restype, f.argtypes = r, a
        L.bound_names = sorted(sig)
        _host = L
    return _host


def host_emu():
    """Build + load the test-only emulator backend (links the lane emulator sources and libmpeghost)."""
    global _emu
    if _emu is None:
        from mpeg_amd import _build
        host()
        srcs = [ROOT / "tests" / "host_emu" / "emu_backend.cpp", ROOT / "tests" / "kernel_emu" / "emu.cpp"]
        deps = srcs + sorted(_build.CSRC.glob("*.h")) + sorted(_build.HOST.glob("*.hpp")) + [_build.LIBMPEGHOST]
        if not (HOST_EMU.exists() and all(d.stat().st_mtime <= HOST_EMU.stat().st_mtime for d in deps)):
            cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DMPG_EMU_CHECKS", "-Wno-unknown-pragmas",
                   "-I", str(_build.INCLUDE), "-I", str(_build.CSRC), "-I", str(_build.HOST), *map(str, srcs), "-o", str(HOST_EMU),
                   "-L", str(_build.LIBMPEGHOST.parent), "-lmpeghost", "-Wl,-rpath," + str(_build.LIBMPEGHOST.parent)]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("host_emu build failed:\n" + r.stdout)
        L = C.CDLL(str(HOST_EMU))
        L.host_emu_video_backend.restype = C.c_void_p
        L.host_emu_video_backend.argtypes = [C.c_int]
        L.host_emu_audio_backend.restype = C.c_void_p
        L.host_emu_audio_backend.argtypes = [C.c_int, C.c_void_p]
        L.host_emu_batch_store.restype = C.c_void_p
        L.host_emu_batch_store.argtypes = []
        L.host_emu_batch_store_staged_commits.restype = C.c_uint64
        L.host_emu_batch_store_staged_commits.argtypes = [C.c_void_p]
        L.host_emu_batch_store_device_pack_stages.restype = C.c_uint64
        L.host_emu_batch_store_device_pack_stages.argtypes = [C.c_void_p]
        L.host_emu_audio_batch_store.restype = C.c_void_p
        L.host_emu_audio_batch_store.argtypes = []
        L.host_emu_configure.restype = None
        L.host_emu_configure.argtypes = [C.c_int, C.c_void_p]
        _emu = L
    return _emu


def frame_planes(f: HostFrame):
    y = np.ctypeslib.as_array(C.cast(f.y, C.POINTER(C.c_uint8)), shape=(f.luma_bytes,)).copy()
    cb = np.ctypeslib.as_array(C.cast(f.cb, C.POINTER(C.c_uint8)), shape=(f.chroma_bytes,)).copy()
    cr = np.ctypeslib.as_array(C.cast(f.cr, C.POINTER(C.c_uint8)), shape=(f.chroma_bytes,)).copy()
    return y, cb, cr


class HostVideo:
    """mpeg.NewVideo + Video.Decode through the product's parser; backend = HIP device or (tests) emulator."""

    def __init__(self, data: bytes, device=None, emu_flavour=None):
        self._data = data
        L = host()
        if device is not None:
            self.h = L.mpeghost_video_open(device, data, len(data))
        else:
            be = host_emu().host_emu_video_backend(emu_flavour or 0)
            self.h = L.mpeghost_video_open_backend(be, data, len(data))
        if not self.h:
            raise RuntimeError(L.mpeghost_last_error().decode())

    def decode(self):
        f = HostFrame()
        rc = host().mpeghost_video_decode(self.h, C.byref(f))
        if rc < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return f if rc == 1 else None

    time = property(lambda s: host().mpeghost_video_time(s.h))
    has_ended = property(lambda s: bool(host().mpeghost_video_has_ended(s.h)))

    def rewind(self):
        host().mpeghost_video_rewind(self.h)

    def set_lookahead(self, on):
        host().mpeghost_video_set_lookahead(self.h, 1 if on else 0)

    def set_host_mirror(self, on):
        host().mpeghost_video_set_host_mirror(self.h, 1 if on else 0)

    def set_device_pack_from(self, n_mbs):
        host().mpeghost_video_set_device_pack_from(self.h, n_mbs)

    def rgba(self, w, h):
        p = host().mpeghost_video_rgba(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h, w, 4)).copy()

    def stats(self):
        out = (C.c_uint64 * 8)()
        host().mpeghost_video_stats(self.h, C.byref(out))
        return dict(zip(("pictures", "submits", "macroblocks", "coded_blocks", "raw_macroblocks", "invalid_blocks",
                         "duplicate_splits", "range_skips"), list(out)))

    def close(self):
        if self.h:
            host().mpeghost_video_close(self.h)
            self.h = None


class HostAudio:
    def __init__(self, data: bytes, device=None, fma=0, fmt=0, window=None):
        self._data = data
        L = host()
        if device is not None:
            self.h = L.mpeghost_audio_open(device, data, len(data), fma, fmt)
        else:
            self._win = np.ascontiguousarray(window, np.float32)
            be = host_emu().host_emu_audio_backend(fma, self._win.ctypes.data)
            self.h = L.mpeghost_audio_open_backend(be, data, len(data), fmt)
        if not self.h:
            raise RuntimeError(L.mpeghost_last_error().decode())
        self.fmt = fmt

    def decode(self):
        t = C.c_double()
        p = host().mpeghost_audio_decode(self.h, C.byref(t))
        if not p:
            return None
        if self.fmt == 3:
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(2304,)).copy()
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(2304,)).copy()

    time = property(lambda s: host().mpeghost_audio_time(s.h))
    has_ended = property(lambda s: bool(host().mpeghost_audio_has_ended(s.h)))

    def rewind(self):
        host().mpeghost_audio_rewind(self.h)

    def set_lookahead(self, on):
        host().mpeghost_audio_set_lookahead(self.h, 1 if on else 0)

    def decode_view(self):
        """-> (pointer value, time) of the decoder's own Samples buffer, or None: no copy (lifetime tests)"""
        t = C.c_double()
        p = host().mpeghost_audio_decode(self.h, C.byref(t))
        return (p, t.value) if p else None

    samplerate = property(lambda s: host().mpeghost_audio_samplerate(s.h))
    channels = property(lambda s: host().mpeghost_audio_channels(s.h))

    def close(self):
        if self.h:
            host().mpeghost_audio_close(self.h)
            self.h = None


PACKET_VIDEO_1, PACKET_AUDIO_1 = 0xE0, 0xC0  # demux.go:11-29


class HostDemux:
    """mpeg.Demux over a whole program stream."""

    def __init__(self, data: bytes):
        self._data = data
        self.h = host().mpeghost_demux_open(data, len(data))
        if not self.h:
            raise RuntimeError(host().mpeghost_last_error().decode())

    def start_time(self, typ):
        return host().mpeghost_demux_start_time(self.h, typ)

    def duration(self, typ):
        return host().mpeghost_demux_duration(self.h, typ)

    def probe(self, size):
        return bool(host().mpeghost_demux_probe(self.h, size))

    def streams(self):
        out = (C.c_int * 2)()
        host().mpeghost_demux_streams(self.h, C.byref(out))
        return out[0], out[1]

    def rewind(self):
        host().mpeghost_demux_rewind(self.h)

    def _packet(self, typ, pts, n, data):
        if typ == 0:
            return None
        return typ, pts.value, C.string_at(data.value, n.value)

    def decode(self):
        pts, n, data = C.c_double(), C.c_size_t(), C.c_void_p()
        return self._packet(host().mpeghost_demux_decode(self.h, C.byref(pts), C.byref(n), C.byref(data)), pts, n, data)

    def seek(self, seconds, typ, force_intra):
        pts, n, data = C.c_double(), C.c_size_t(), C.c_void_p()
        return self._packet(host().mpeghost_demux_seek(self.h, seconds, typ, int(force_intra), C.byref(pts), C.byref(n), C.byref(data)),
                            pts, n, data)

    def close(self):
        if self.h:
            host().mpeghost_demux_close(self.h)
            self.h = None


class HostMpeg:
    """mpeg.MPEG over a whole program stream; device=None runs the decoders on the test-only lane emulator."""

    def __init__(self, data: bytes, device=None, window=None, flavour=0):
        self._data = data
        L = host()
        if device is not None:
            self.h = L.mpeghost_mpeg_open(device, data, len(data))
        else:
            E = host_emu()
            self._win = np.ascontiguousarray(window, np.float32)
            E.host_emu_configure(flavour, self._win.ctypes.data)
            mk_v = C.cast(E.host_emu_make_video, C.c_void_p)
            mk_a = C.cast(E.host_emu_make_audio, C.c_void_p)
            self.h = L.mpeghost_mpeg_open_backends(mk_v, mk_a, data, len(data))
        if not self.h:
            raise RuntimeError(L.mpeghost_last_error().decode())

    def info(self):
        out = (C.c_int * 6)()
        host().mpeghost_mpeg_info(self.h, C.byref(out))
        return dict(zip(("video_streams", "audio_streams", "width", "height", "samplerate", "channels"), out))

    def _call(self, name, *a):
        return getattr(host(), "mpeghost_mpeg_" + name)(self.h, *a)

    def probe(self, size):
        return self._call("probe", size) == 1

    def has_headers(self):
        return self._call("has_headers") == 1

    duration = property(lambda s: s._call("duration"))
    time = property(lambda s: s._call("time"))
    audio_time = property(lambda s: s._call("audio_time"))
    video_time = property(lambda s: s._call("video_time"))
    framerate = property(lambda s: s._call("framerate"))
    has_ended = property(lambda s: s._call("has_ended") == 1)

    def set_enabled(self, video, audio):
        self._call("set_enabled", int(video), int(audio))

    def get_enabled(self):
        out = (C.c_int * 2)()
        self._call("get_enabled", C.byref(out))
        return bool(out[0]), bool(out[1])

    def set_audio_stream(self, index):
        self._call("set_audio_stream", int(index))

    def set_loop(self, loop):
        self._call("set_loop", int(loop))

    loop = property(lambda s: s._call("loop") == 1)

    def rewind(self):
        self._call("rewind")

    def count_callbacks(self, video=True, audio=True):
        self._call("count_callbacks", int(video), int(audio))

    def callback_counts(self):
        out = (C.c_int * 2)()
        self._call("callback_counts", C.byref(out))
        return out[0], out[1]

    def decode(self, tick):
        self._call("decode", float(tick))

    def decode_video(self):
        f = HostFrame()
        return f if self._call("decode_video", C.byref(f)) == 1 else None

    def decode_audio(self):
        t = C.c_double()
        p = self._call("decode_audio", C.byref(t))
        if not p:
            return None
        return t.value, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(2304,)).copy()

    def seek(self, seconds, exact):
        return self._call("seek", float(seconds), int(exact)) == 1

    def seek_frame(self, seconds, exact):
        f = HostFrame()
        return f if self._call("seek_frame", float(seconds), int(exact), C.byref(f)) == 1 else None

    def close(self):
        if self.h:
            host().mpeghost_mpeg_close(self.h)
            self.h = None


class HostBatch:
    """mpeg::VideoBatch: n elementary streams of one picture size, one device call per decode_all()."""

    def __init__(self, n_streams: int, device=None, threads: int = 1):
        L = host()
        if device is not None:
            self.h = L.mpeghost_batch_open(device, n_streams)
        else:
            self.emu_store = host_emu().host_emu_batch_store()
            self.h = L.mpeghost_batch_open_store(self.emu_store, n_streams)
        if not self.h:
            raise RuntimeError(L.mpeghost_last_error().decode())
        if threads > 1:
            L.mpeghost_batch_set_threads(self.h, threads)
        self._keep = []

    def add_stream(self, data: bytes) -> int:
        self._keep.append(data)
        i = host().mpeghost_batch_add_stream(self.h, data, len(data))
        if i < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return i

    def decode_all(self, fetch=True) -> int:
        n = host().mpeghost_batch_decode_all(self.h, int(fetch))
        if n < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return n

    def frame(self, stream: int):
        f = HostFrame()
        return f if host().mpeghost_batch_frame(self.h, stream, C.byref(f)) == 1 else None

    def set_device_pack(self, on: bool):
        """Staged submits of sparse pictures validated and packed on the host (default) or on the device (errors deferred)."""
        host().mpeghost_batch_set_device_pack(self.h, int(on))

    def sync(self):
        if host().mpeghost_batch_sync(self.h) != 0:
            raise RuntimeError(host().mpeghost_last_error().decode())

    device_pack = property(lambda s: bool(host().mpeghost_batch_device_pack(s.h)))

    def refused_streams(self):
        out = (C.c_uint32 * 1024)()
        n = host().mpeghost_batch_refused_streams(self.h, out, 1024)
        return list(out[:min(n, 1024)])

    def damage_next_picture(self, stream: int):
        host().mpeghost_batch_debug_damage_next_picture(self.h, stream)

    def counters(self):
        out = (C.c_uint64 * 2)()
        host().mpeghost_batch_counters(self.h, C.byref(out))
        c = {"device_submits": out[0], "queued_pictures": out[1]}
        if getattr(self, "emu_store", None):
            c["staged_commits"] = int(host_emu().host_emu_batch_store_staged_commits(self.emu_store))
            c["device_pack_stages"] = int(host_emu().host_emu_batch_store_device_pack_stages(self.emu_store))
        return c

    def close(self):
        if self.h:
            host().mpeghost_batch_close(self.h)
            self.h = None


class HostSharded:
    """mpeg::ShardedVideoBatch: streams sharded over several devices (stream s -> device s mod G), one host thread and
    one VideoBatch per device.  devices: a list of mpeghost devices, or an int = that many test-only emulator stores."""

    def __init__(self, n_streams: int, devices):
        L = host()
        if isinstance(devices, int):
            E = host_emu()
            arr = (C.c_void_p * devices)(*[E.host_emu_batch_store() for _ in range(devices)])
            self.h = L.mpeghost_sharded_open_stores(arr, devices, n_streams)
            self.shards = devices
        else:
            arr = (C.c_void_p * len(devices))(*devices)
            self.h = L.mpeghost_sharded_open(arr, len(devices), n_streams)
            self.shards = len(devices)
        if not self.h:
            raise RuntimeError(L.mpeghost_last_error().decode())
        self._keep = []

    def add_stream(self, data: bytes) -> int:
        self._keep.append(data)
        i = host().mpeghost_sharded_add_stream(self.h, data, len(data))
        if i < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return i

    def decode_all(self, fetch=True) -> int:
        n = host().mpeghost_sharded_decode_all(self.h, int(fetch))
        if n < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return n

    def frame(self, stream: int):
        f = HostFrame()
        return f if host().mpeghost_sharded_frame(self.h, stream, C.byref(f)) == 1 else None

    def set_threads(self, n: int):
        host().mpeghost_sharded_set_threads(self.h, n)

    def device_of(self, stream: int) -> int:
        return host().mpeghost_sharded_device_of(self.h, stream)

    def set_device_pack(self, on: bool):
        host().mpeghost_sharded_set_device_pack(self.h, int(on))

    def sync(self):
        if host().mpeghost_sharded_sync(self.h) != 0:
            raise RuntimeError(host().mpeghost_last_error().decode())

    def counters(self, shard: int):
        out = (C.c_uint64 * 2)()
        host().mpeghost_sharded_counters(self.h, shard, C.byref(out))
        return {"device_submits": out[0], "queued_pictures": out[1]}

    def close(self):
        if self.h:
            host().mpeghost_sharded_close(self.h)
            self.h = None


class HostAudioBatch:
    """mpeg::AudioBatch: n MP2 streams, one synthesis call per decode_all().  fmt: 0 F32N, 1 F32NLR, 2 F32, 3 S16."""

    def __init__(self, n_streams: int, device=None, fmt=0, fma=0, window=None):
        L = host()
        self.fmt = fmt
        if device is not None:
            self.h = L.mpeghost_audio_batch_open(device, n_streams, fmt, fma)
        else:
            E = host_emu()
            self._win = np.ascontiguousarray(window, np.float32)
            E.host_emu_configure(0, self._win.ctypes.data)
            self.h = L.mpeghost_audio_batch_open_store(E.host_emu_audio_batch_store(), n_streams, fmt, fma)
        if not self.h:
            raise RuntimeError(L.mpeghost_last_error().decode())
        self._keep = []

    def add_stream(self, data: bytes) -> int:
        self._keep.append(data)
        i = host().mpeghost_audio_batch_add_stream(self.h, data, len(data))
        if i < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return i

    def decode_all(self) -> int:
        n = host().mpeghost_audio_batch_decode_all(self.h)
        if n < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return n

    def samples(self, stream: int):
        """The stream's frame as the reference's golden test hashes it: 2304 (or 1152 + 1152) elements."""
        t, right = C.c_double(), C.c_void_p()
        p = host().mpeghost_audio_batch_samples(self.h, stream, C.byref(t), C.byref(right))
        if not p:
            return None
        if self.fmt == 3:
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(2304,)).copy()
        if self.fmt == 1:
            l = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(1152,))
            r = np.ctypeslib.as_array(C.cast(right.value, C.POINTER(C.c_float)), shape=(1152,))
            return np.concatenate([l, r])
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(2304,)).copy()

    def decode_stream_directly(self, stream: int) -> bool:
        r = host().mpeghost_audio_batch_decode_stream(self.h, stream)
        if r < 0:
            raise RuntimeError(host().mpeghost_last_error().decode())
        return r == 1

    device_calls = property(lambda s: host().mpeghost_audio_batch_device_calls(s.h))

    def close(self):
        if self.h:
            host().mpeghost_audio_batch_close(self.h)
            self.h = None
