#!/usr/bin/env python3
"""kNN-only timing on synthetic encodings in HBM (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cellcomm_amd import _lib
from cellcomm_amd.knn import knn_graph_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
nq = int(sys.argv[4]) if len(sys.argv) > 4 else n
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
_lib.ensure_init(0)
if os.environ.get('CC_FULL'):
    _lib.lib().cc_knn_force_full_select(1)   # route 1a whatever the size
if os.environ.get('CC_KNN_COLLECT_SPLITS'):  # sweep knob (tools/knn_split_sweep.sh): an option of the workspace, not an env read in the library
    _lib.check(_lib.lib().cc_knn_set_option(None, 5, int(os.environ['CC_KNN_COLLECT_SPLITS'])))
if os.environ.get('CC_KNN_SAMPLE_SPLITS'):
    _lib.check(_lib.lib().cc_knn_set_option(None, 7, int(os.environ['CC_KNN_SAMPLE_SPLITS'])))
if os.environ.get('CC_KNN_SAMPLE_STRIDE'):
    _lib.check(_lib.lib().cc_knn_set_option(None, 8, int(os.environ['CC_KNN_SAMPLE_STRIDE'])))
if os.environ.get('CC_KNN_SAMPLE_GROUPS'):
    _lib.check(_lib.lib().cc_knn_set_option(None, 9, int(os.environ['CC_KNN_SAMPLE_GROUPS'])))
if os.environ.get('CC_KNN_COLLECT_GROUPS'):
    _lib.check(_lib.lib().cc_knn_set_option(None, 10, int(os.environ['CC_KNN_COLLECT_GROUPS'])))
if os.environ.get('CC_KNN_COLLECT_NORM_FORM'):
    _lib.check(_lib.lib().cc_knn_set_option(None, 11, int(os.environ['CC_KNN_COLLECT_NORM_FORM'])))
torch.manual_seed(20260105)   # the same points in every process: A/B runs compare like with like
x = torch.rand((n, d), device='cuda'); torch.cuda.synchronize()
knn_graph_device(x.data_ptr(), n, d, k, 0, nq)
_lib.profile_enable(True); _lib.profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    knn_graph_device(x.data_ptr(), n, d, k, 0, nq)
t1 = time.perf_counter()
p = _lib.profile_snapshot()
line_host = ''
if nq == n and os.environ.get('CC_KNN_HOST_LEG', '1') != '0':   # the same search from HOST points (cc_knn_l2, what cc.knn_graph(ndarray) calls)
    import cellcomm_amd as cc
    xh = x.cpu().numpy()
    cc.knn_graph(xh, k)
    th = time.perf_counter()
    for _ in range(reps):
        cc.knn_graph(xh, k)
    line_host = f' (host points: {1e3*(time.perf_counter()-th)/reps:.2f} ms/call)'
print(f'n={n} d={d} k={k} nq={nq}: {1e3*(t1-t0)/reps:.2f} ms/call' + line_host, {kk: round(v['ms']/reps, 3) for kk, v in p.items() if v['launches']},
      'fallback', _lib.lib().cc_knn_last_fallback_count())
