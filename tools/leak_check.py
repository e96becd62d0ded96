#!/usr/bin/env python3
"""Device-memory leak check: create/use/free every handle type in a loop and watch hipMemGetInfo."""
import os, sys, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth, pipeline
from cellcomm_amd.generator import build_cont_generator

_lib.ensure_init(0)

def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20

def cycle(i):
    m = synth.synth_csr(300 + i % 7, 2000, density=0.05, seed=i)
    for dtype in ('f32', 'f16x2'):
        enc = cc.HipEncoder(16, 2000, 'cont' if i % 2 else 'cls', seed=i, dtype=dtype, batch_rows=128)
        z = enc.predict(m)
        res = pipeline.encode_and_knn(enc, m, 5)
        del enc, res
    cc.knn_graph(z, 7)
    gen = build_cont_generator(16, 2000)
    gen.predict((np.random.rand(8, 16).astype(np.float32), np.random.rand(8, 16).astype(np.float32)))
    del gen, m
    gc.collect()

for i in range(3):
    cycle(i)                       # warm the grow-only scratch buffers
base = free_mb()
for i in range(3, 63):
    cycle(i)
    if i % 20 == 2:
        print(f'after {i - 2} cycles: free {free_mb():.1f} MiB (start {base:.1f})', flush=True)
end = free_mb()
print('delta MiB over 60 cycles:', round(base - end, 1))
sys.exit(1 if base - end > 64 else 0)
