#!/usr/bin/env python3
"""Probe (round 3): cc_encode_dense on 16384 pinned host rows at G = 30000, a few calls; run under
rocprofv3 --kernel-trace --memory-copy-trace to see how the copies of pass i+1 sit against the kernels of pass i."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cellcomm_amd as cc

g, n = 30000, 16384
enc = cc.HipEncoder(64, g, 'cont', seed=1)
x = torch.empty((n, g), dtype=torch.float32).pin_memory()
x.uniform_(0, 3).floor_()
xn = x.numpy()
enc.predict(xn[:4096])
for _ in range(3):
    t0 = time.perf_counter()
    enc.predict(xn)
    dt = time.perf_counter() - t0
    print(f'{n / dt:,.0f} cells/s ({dt * 1e3:.1f} ms)', flush=True)
