"""kNN graph on latent encodings (K5) — host side of cc_knn_l2 / cc_knn_l2_dev.

The reference builds neighbour graphs only inside umap.UMAP / sklearn TSNE
(reference src/intercepts/plot_intercepts.py:92,176,183); this is the exact,
brute-force replacement the north star asks for.  Semantics: oracle/knn_oracle.c.
"""
import numpy as np

from . import _lib


def knn_graph(encodings, k, include_self=False):
    """(idx int32 [n x k], dist float32 [n x k]) — for every row the k nearest
    other rows under squared L2, ordered by (distance, index).  `encodings` is a
    host array-like [n x d]."""
    lib = _lib.ensure_init()
    x = np.ascontiguousarray(encodings, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError(f'encodings must be 2-D, got shape {x.shape}')
    n, d = x.shape
    idx = np.empty((n, k), dtype=np.int32)
    dist = np.empty((n, k), dtype=np.float32)
    _lib.check(lib.cc_knn_l2(x.ctypes.data, n, d, int(k), int(bool(include_self)), idx.ctypes.data,
                             dist.ctypes.data))
    return idx, dist


def knn_graph_device(dev_ptr, n_total, d, k, q_begin=0, n_q=None, include_self=False, out=None):
    """Same for encodings already in HBM (`dev_ptr` = device address of a
    contiguous float32 [n_total x d]); queries are rows [q_begin, q_begin+n_q) and
    the returned indices are global row numbers.  `out=(idx, dist)` supplies the
    host result arrays (e.g. views of pinned memory, which makes the copy-out DMA)."""
    lib = _lib.ensure_init()
    n_q = n_total - q_begin if n_q is None else n_q
    if out is None:
        idx = np.empty((n_q, k), dtype=np.int32)
        dist = np.empty((n_q, k), dtype=np.float32)
    else:
        idx, dist = out
        if idx.shape != (n_q, k) or dist.shape != (n_q, k) or idx.dtype != np.int32 or dist.dtype != np.float32 \
                or not idx.flags.c_contiguous or not dist.flags.c_contiguous:
            raise ValueError(f'out arrays must be C-contiguous int32/float32 of shape {(n_q, k)}')
    _lib.check(lib.cc_knn_l2_dev(dev_ptr, n_total, d, q_begin, n_q, int(k), int(bool(include_self)),
                                 idx.ctypes.data, dist.ctypes.data))
    return idx, dist


def last_fallback_count():
    return int(_lib.lib().cc_knn_last_fallback_count())


def force_fallback(on):
    _lib.check(_lib.lib().cc_knn_force_fallback(1 if on else 0))


def force_full_select(on):
    """Test hook: route every call through the full running-threshold select (stage 1a)."""
    _lib.check(_lib.lib().cc_knn_force_full_select(1 if on else 0))


def use_fp16_collect(on):
    """Stage 1b collect pass on the fp16 (default) or fp32 matrix cores; same results, different speed."""
    _lib.check(_lib.lib().cc_knn_use_fp16_collect(1 if on else 0))


def use_coarse_collect(on):
    """Stage 1b collect as ONE fp16 product per pair + direct fp32 re-evaluation of the parked pairs (default), or
    every pair on the fp32 matrix cores (the cross-check route); same results, different speed
    (CC_KNN_OPT_COARSE_COLLECT)."""
    _lib.check(_lib.lib().cc_knn_set_option(None, 6, 1 if on else 0))


def use_fp16_sample(on):
    """Stage 1b sample pass as slot minima on the fp16 matrix cores (default) or as list select on the fp32 ones;
    same results, different threshold and speed (CC_KNN_OPT_FP16_SELECT)."""
    _lib.check(_lib.lib().cc_knn_set_option(None, 3, 1 if on else 0))


def set_sample_layout(splits=0, stride=0, groups=0):
    """Tuning / test hook: candidate ranges of the fp16 sample pass, the sampling stride in candidate blocks and the
    32-query groups per wave (0 = chosen by the library).  Results never depend on them
    (CC_KNN_OPT_SAMPLE_SPLITS / _STRIDE / _GROUPS)."""
    _lib.check(_lib.lib().cc_knn_set_option(None, 7, int(splits)))
    _lib.check(_lib.lib().cc_knn_set_option(None, 8, int(stride)))
    _lib.check(_lib.lib().cc_knn_set_option(None, 9, int(groups)))


def set_collect_groups(groups=0):
    """Tuning / test hook: 32-query groups per wave of the coarse collect pass (1 or 2; 0 = chosen by the
    library).  Results never depend on it (CC_KNN_OPT_COLLECT_GROUPS)."""
    _lib.check(_lib.lib().cc_knn_set_option(None, 10, int(groups)))


def set_collect_norm_form(form=0):
    """Tuning / test hook: how the coarse collect's tile takes the two norms — 1 = a K-step of their own (ND + 1
    MFMAs, one-instruction sign test), 2 = through the accumulator (ND MFMAs, the kernel of the rare-hit shapes);
    0 = chosen by the library.  Results never depend on it (CC_KNN_OPT_COLLECT_NORM_FORM)."""
    _lib.check(_lib.lib().cc_knn_set_option(None, 11, int(form)))


def use_fp64_prefilter(on):
    """Stage 3: narrow each uncertified query's candidates with a float64 matrix-core pass (default) before
    the exact evaluation, or scan all candidates; same results, different speed."""
    _lib.check(_lib.lib().cc_knn_use_fp64_prefilter(1 if on else 0))


def last_sample_stride():
    """Every n-th candidate block fed the threshold of the last search (0: it ran without a sample pass)."""
    import ctypes
    out = ctypes.c_int64()
    _lib.check(_lib.lib().cc_knn_stat(None, 2, ctypes.byref(out)))
    return int(out.value)


def last_collect_norm_form():
    """Tile of the last search's coarse collect: 1 = norm K-step, 2 = norms through the accumulator, 0 = no coarse
    collect ran (small inputs, fp32 routes)."""
    import ctypes
    out = ctypes.c_int64()
    _lib.check(_lib.lib().cc_knn_stat(None, 3, ctypes.byref(out)))
    return int(out.value)


def last_full_scan_count():
    """Uncertified queries of the last call that had to scan every candidate in float64."""
    return int(_lib.lib().cc_knn_last_full_scan_count())
