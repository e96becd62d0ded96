#!/usr/bin/env python3
"""Wall-clock breakdown of one bench step (host side) — diagnostic only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, pipeline, synth
from cellcomm_amd.knn import knn_graph_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
_lib.ensure_init(0)
m, keep = synth.synth_csr_device(n, 30000, seed=1234)
enc = cc.HipEncoder(64, 30000, 'cont', seed=4321)
out = torch.empty((n, 64), device='cuda')
host = np.empty((n, 64), dtype=np.float32)
pinned = torch.empty((n, 64), pin_memory=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.predict_to_device(m, out.data_ptr(), 0, n); t1 = time.perf_counter()
    idx, dist = knn_graph_device(out.data_ptr(), n, 64, 15); t2 = time.perf_counter()
    host[:] = out.cpu().numpy(); t3 = time.perf_counter()
    pinned.copy_(out, non_blocking=True); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f'it{it}: encode {1e3*(t1-t0):.1f} ms | knn {1e3*(t2-t1):.1f} ms | enc D2H pageable {1e3*(t3-t2):.1f} ms | pinned {1e3*(t4-t3):.1f} ms')
_lib.profile_enable(True); _lib.profile_reset()
t0 = time.perf_counter(); idx, dist = knn_graph_device(out.data_ptr(), n, 64, 15); t1 = time.perf_counter()
print('knn with profiling', 1e3*(t1-t0), {k: round(v['ms'], 2) for k, v in _lib.profile_snapshot().items() if v['launches']})
x = torch.rand((n, 64), device='cuda'); torch.cuda.synchronize()
_lib.profile_reset()
t0 = time.perf_counter(); idx, dist = knn_graph_device(x.data_ptr(), n, 64, 15); t1 = time.perf_counter()
print('knn uniform data   ', 1e3*(t1-t0), {k: round(v['ms'], 2) for k, v in _lib.profile_snapshot().items() if v['launches']}, 'fallback', _lib.lib().cc_knn_last_fallback_count())
