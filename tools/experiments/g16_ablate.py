#!/usr/bin/env python3
"""Encoder-only timing of the fp16 layers (no kNN: safe for ablation builds whose results are garbage).
usage: g16_ablate.py <cells> <genes> <dtype> [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth

cells, genes, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
_lib.ensure_init(0)
m, _keep = synth.synth_csr_device(cells, genes, density=0.05, seed=1234)
enc = cc.HipEncoder(64, genes, 'cont', seed=4321, dtype=dtype)
out = torch.empty((cells, 64), dtype=torch.float32, device='cuda')
enc.predict_to_device(m, out.data_ptr(), 0, cells)
_lib.profile_enable(True)
_lib.profile_reset()
for _ in range(reps):
    enc.predict_to_device(m, out.data_ptr(), 0, cells)
snap = _lib.profile_snapshot()
for k in ('gemm_l1', 'gemm_l2'):
    v = snap[k]
    print(f'{k}: {v["ms"] / v["launches"]:.3f} ms/launch, {v["work"] / (v["ms"] * 1e-3) / 1e12:.1f} TFLOP/s', end='  |  ')
print()
