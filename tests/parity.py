"""Shared comparison helpers for the parity tests."""
from __future__ import annotations

import numpy as np


def assert_planes_equal(ref, got, what=""):
    for name, a, b in zip(("Y", "Cb", "Cr"), ref, got):
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError("%s plane %s: %d bytes differ, first at %d (want %d got %d)" %
                                 (what, name, len(bad), bad[0], a[bad[0]], b[bad[0]]))


def run_and_compare(ref_store, dut_store, seq, check_rgba=False, stream=0):
    """Feed every Submit to both stores; after each picture all three slots must be byte-equal."""
    for i, s in enumerate(seq):
        ref_store.submit(s.pics, s.mbs, s.coefs)
        dut_store.submit(s.pics, s.mbs, s.coefs)
        for slot in range(3):
            assert_planes_equal(ref_store.read_planes(stream, slot), dut_store.read_planes(stream, slot),
                                "picture %d (type %d) slot %d" % (i, s.picture_type, slot))
        if check_rgba:
            a, b = ref_store.read_rgba(stream, s.cur), dut_store.read_rgba(stream, s.cur)
            if not np.array_equal(a, b):
                bad = np.argwhere(a != b)
                raise AssertionError("picture %d RGBA: %d bytes differ, first at %s" % (i, len(bad), bad[0]))


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def mirror_ring(X):
    """Audio.v contents from 32 DCT outputs per slot: X[2, 16, 32] -> v[2, 1024] (audio.go:708-771:
    d[48-k] = d[48+k] = -X[k], d[k-16] = X[k], d[16] = 0)."""
    X = np.asarray(X, np.float32)
    v = np.zeros((2, 16, 64), np.float32)
    for k in range(32):
        if k <= 16:
            v[:, :, 48 - k] = -X[:, :, k]
        if 1 <= k <= 15:
            v[:, :, 48 + k] = -X[:, :, k]
        if k >= 17:
            v[:, :, 48 - k] = -X[:, :, k]
            v[:, :, k - 16] = X[:, :, k]
        if k == 16:
            v[:, :, 0] = X[:, :, k]
    return v.reshape(2, 1024)
