#!/usr/bin/env python3
"""Probe (round 4): s_memtime stamps inside the collect's stage loop (a library built from csrc/knn.hip +
tools/experiments/knn_stamps.patch, swapped in by ab_lib.sh): per stage of waves 0 and 2 of two workgroups, the ticks
between the stamps.   python3 tools/experiments/knn_stamps.py <n> <nq> <k>"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cellcomm_amd import _lib
from cellcomm_amd.knn import knn_graph_device
lib = _lib.ensure_init(0)
n, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.rand((n, 64), device='cuda'); torch.cuda.synchronize()
for _ in range(2):
    knn_graph_device(x.data_ptr(), n, 64, k, 0, nq)
buf = (ctypes.c_ulonglong * (4 * 64 * 8))()
lib.cc_debug_knn_stamps.argtypes = [ctypes.c_void_p]
assert lib.cc_debug_knn_stamps(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(4, 64, 8)
names = ['DMA issue', 'fragment reads + lgkmcnt(0)', 'plain test + MFMA issue', 'sign masks (waits for the MFMAs)', 'park', 'vmcnt wait', 'barrier']
for w in range(4):
    tw = t[w]
    ok = tw[:, 0] > 0
    if ok.sum() < 10:
        print(f'slot {w}: no stamps'); continue
    tw = tw[ok]
    d = np.diff(tw, axis=1)
    per = np.diff(tw[:, 0])
    per = per[per > 0]
    print(f'workgroup {w // 2} wave {2 * (w % 2)}: {ok.sum()} stages, period {np.median(per):.0f} ticks (p10 {np.percentile(per, 10):.0f}, p90 {np.percentile(per, 90):.0f})')
    for i, nm in enumerate(names):
        print(f'   {nm:36s} median {np.median(d[:, i]):7.0f}  p10 {np.percentile(d[:, i], 10):7.0f}  p90 {np.percentile(d[:, i], 90):7.0f}')
