"""Oracle: exact brute-force kNN (float64).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED — the reference contains no kNN code; semantics are
defined in oracle/knn_oracle.c.  ``knn`` calls the C restatement (built by
oracle/Makefile); ``knn_numpy`` is an independent NumPy statement of the same
arithmetic used to validate the C build on small inputs.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} missing: run `make -C oracle` (or __graft_entry__.build())')
        _lib = ctypes.CDLL(path)
        _lib.knn_oracle.restype = ctypes.c_int
        _lib.knn_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                    ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p]
        _lib.encoder_oracle_forward.restype = ctypes.c_int
        _lib.encoder_oracle_forward.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_void_p]
    return _lib


def knn(x, k, include_self=False, q_begin=0, n_q=None, return_dist64=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    n_q = n - q_begin if n_q is None else n_q
    idx = np.empty((n_q, k), dtype=np.int32)
    dist = np.empty((n_q, k), dtype=np.float32)
    d64 = np.empty((n_q, k), dtype=np.float64)
    rc = _load().knn_oracle(x.ctypes.data, n, d, q_begin, n_q, k, int(include_self),
                            idx.ctypes.data, dist.ctypes.data, d64.ctypes.data)
    if rc != 0:
        raise ValueError(f'Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, '
                         f'n_samples_fit = {n - (0 if include_self else 1)}')
    return (idx, dist, d64) if return_dist64 else (idx, dist)


def knn_numpy(x, k, include_self=False):
    """Same arithmetic in NumPy: sequential float64 accumulation over dimensions,
    separate multiply and add (NumPy never fuses them), order (dist, idx)."""
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x32.shape
    x64 = x32.astype(np.float64)
    idx = np.empty((n, k), dtype=np.int32)
    dist = np.empty((n, k), dtype=np.float32)
    for i in range(n):
        acc = np.zeros(n, dtype=np.float64)
        for t in range(d):
            diff = x64[i, t] - x64[:, t]
            acc = acc + diff * diff
        cand = np.arange(n)
        if not include_self:
            keep = cand != i
            cand, acc = cand[keep], acc[keep]
        order = np.lexsort((cand, acc))[:k]
        idx[i] = cand[order]
        dist[i] = acc[order].astype(np.float32)
    return idx, dist


def encoder_forward_c(arch, weights, x, encoding_size):
    """Scalar float32 C port of the encoder forward (oracle/encoder_oracle.c)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, g = x.shape
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    ptrs = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    out = np.empty((n, encoding_size), dtype=np.float32)
    _load().encoder_oracle_forward(0 if arch == 'cont' else 1, g, encoding_size, ptrs, x.ctypes.data, n,
                                   out.ctypes.data)
    return out
