#!/usr/bin/env python3
"""Latency of encoder.predict for small batches (the reference samples minibatches of 128 cells)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth

_lib.ensure_init(0)
g = 30000
m, _ = synth.synth_csr_device(4096, g, seed=3)
only_ll = len(sys.argv) > 1 and sys.argv[1] == 'low_latency'      # just the split-K form (for a kernel trace of it)
for dtype, low_latency in ((('f32', True),) if only_ll else (('f32', False), ('f32', True), ('f16x2', False))):
    enc = cc.HipEncoder(64, g, 'cont', seed=1, dtype=dtype, low_latency=low_latency)
    out = torch.empty((4096, 64), device='cuda')
    for n in ((1, 32, 128) if low_latency else (1, 32, 128, 512, 2048, 4096)):
        enc.predict_to_device(m, out.data_ptr(), 0, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            enc.predict_to_device(m, out.data_ptr(), 0, n)
        torch.cuda.synchronize()
        print(f'{dtype}{" low_latency" if low_latency else ""} n={n:5d}: {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms')
