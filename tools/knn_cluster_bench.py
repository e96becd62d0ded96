#!/usr/bin/env python3
"""kNN timing / fallback count on index-correlated data (cells sorted by cluster) — the case where the
block-wise 1/SAMPLE sample of the threshold pass is least representative.  Diagnostic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cellcomm_amd import _lib
from cellcomm_amd.knn import knn_graph_device

_lib.ensure_init(0)
n, d, k = 100000, 64, 15
rng = np.random.default_rng(5)
for csize, spread in ((600, 0.01), (40, 0.01), (40, 0.3), (8, 0.01)):
    nc = n // csize
    centres = rng.random((nc, d), dtype=np.float32) * 10
    lab = np.sort(rng.integers(0, nc, n))
    x = (centres[lab] + rng.standard_normal((n, d)).astype(np.float32) * spread).astype(np.float32)
    t = torch.from_numpy(x).cuda(); torch.cuda.synchronize()
    knn_graph_device(t.data_ptr(), n, d, k)
    _lib.profile_enable(True); _lib.profile_reset()
    t0 = time.perf_counter(); knn_graph_device(t.data_ptr(), n, d, k); t1 = time.perf_counter()
    p = _lib.profile_snapshot()
    print(f'clusters of ~{csize} (sd {spread}): {1e3 * (t1 - t0):.2f} ms', {kk: round(v['ms'], 2) for kk, v in p.items() if v['launches']},
          'fallback', _lib.lib().cc_knn_last_fallback_count())
    _lib.profile_enable(False)
