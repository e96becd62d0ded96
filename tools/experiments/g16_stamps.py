#!/usr/bin/env python3
"""Probe (round 3): s_memtime stamps inside the one-plane fp16 GEMM's K loop (build gemm16.hip with -DCC_G16_STAMP, patch in
tools/experiments/g16_stamps.patch): per K-step of waves 0 and 4 of one workgroup, the ticks between the stamps."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth
lib = _lib.ensure_init(0)
n, g = 32768, 30000
m, _ = synth.synth_csr_device(n, g, seed=5)
enc = cc.HipEncoder(64, g, 'cont', seed=6, dtype='f16')
out = torch.empty((n, 64), device='cuda')
for _ in range(3):
    enc.predict_to_device(m, out.data_ptr(), 0, n)
buf = (ctypes.c_ulonglong * (2 * 96 * 8))()
lib.cc_debug_g16_stamps.argtypes = [ctypes.c_void_p]
assert lib.cc_debug_g16_stamps(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(2, 96, 8)
names = ['R1 + DMA issue', 'R2 + wait(4)', 'mfma hc0', 'R3 + wait(8) + mfma hc1', 'R4 + wait(4) + mfma hc2', 'wait(0) + mfma hc3', 'vmcnt + barrier']
for w in (0, 1):
    d = np.diff(t[w], axis=1)[8:88]
    per = (t[w][9:89, 0] - t[w][8:88, 0])
    print(f'wave {4 * w}: K-step period {per.mean():.0f} ticks (min {per.min()}, max {per.max()})')
    for i, nm in enumerate(names):
        print(f'   {nm:28s} mean {d[:, i].mean():7.1f}  p10 {np.percentile(d[:, i], 10):7.1f}  p90 {np.percentile(d[:, i], 90):7.1f}')
print('wave 4 minus wave 0 at stamp 0:', (t[1][8:88, 0] - t[0][8:88, 0]).mean())
