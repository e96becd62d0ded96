#!/usr/bin/env python3
"""What one rank of a P-GPU run does per step, measured on one GPU: encode its 100k/P cells, then search
its queries against 100k candidates (the other ranks' encodings are stand-ins already in HBM).  Everything
but the all-gather itself, so N=1 time / this time is the scaling the collective can only lower.

  python3 tools/shard_step.py <mode> [out.json]     mode: f32 | sparse | f16x2 | f16; the JSON file is appended to
                                                    (one key per mode), e.g. profiles/r03_shard_step.json"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth
from cellcomm_amd.knn import knn_graph_device

_lib.ensure_init(0)
N, G, E, K = 100000, 30000, 64, 15
mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'          # f32 | sparse | f16x2 | f16
enc = cc.HipEncoder(E, G, 'cont', seed=4321, sparse=(mode == 'sparse'), dtype='f32' if mode in ('f32', 'sparse') else mode)
print('mode', mode, flush=True)
full = torch.sigmoid(torch.randn((N, E), device='cuda'))
base = None
rows = []
for P in (tuple(int(v) for v in os.environ['SHARD_P'].split(',')) if os.environ.get('SHARD_P') else (1, 2, 4, 8)):
    n = -(-N // P)
    m, _ = synth.synth_csr_device(n, G, seed=1234)
    pinned_enc = torch.empty((n, E), pin_memory=True)
    idx = np.empty((n, K), np.int32); dist = np.empty((n, K), np.float32)
    def step():
        enc.predict_to_device(m, full.data_ptr(), 0, n)          # this rank's block = rows [0, n)
        pinned_enc.copy_(full[:n], non_blocking=True)
        knn_graph_device(full.data_ptr(), N, E, K, 0, n, out=(idx, dist))
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()                                                       # includes the one-time sparse stream build of this matrix
    first = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    base = base or ms
    print(f'P={P}: {n} cells per rank, {ms:.2f} ms per step -> {N / ms * 1e3:,.0f} cells/s if the all-gather were free, x{base / ms:.2f}  (first step {first:.1f} ms)', flush=True)
    rows.append({'P': P, 'cells_per_rank': n, 'ms_per_step': round(ms, 3), 'speedup_vs_P1': round(base / ms, 3),
                 'first_step_ms': round(first, 2)})
    del m
if len(sys.argv) > 2:
    path = sys.argv[2]
    doc = json.load(open(path)) if os.path.exists(path) else {
        'made_by': 'tools/shard_step.py', 'what': 'per-rank work of a P-GPU step of configs[2] measured on ONE GPU: encode '
        '100k/P cells, D2H of the encodings, kNN of those queries against all 100k candidates; everything but the all-gather'}
    doc[mode] = rows
    json.dump(doc, open(path, 'w'), indent=1)
