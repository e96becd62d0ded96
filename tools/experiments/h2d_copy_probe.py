#!/usr/bin/env python3
"""Probe (round 3): host -> device copy rate of one dense-host pass (4096 rows x 30000 fp32) as cc_encode_dense does it
(hipMemcpy2DAsync into rows of pitch 30016) against one linear copy, from pinned and pageable memory."""
import time
import numpy as np
import torch

rows, g, ld = 4096, 30000, 30016
pinned = torch.empty((rows, g), dtype=torch.float32).pin_memory()
pinned.uniform_()
pageable = pinned.numpy().copy()
dev2d = torch.empty((rows, ld), dtype=torch.float32, device='cuda')
dev1d = torch.empty((rows, g), dtype=torch.float32, device='cuda')


def rate(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return rows * g * 4 * reps / (time.perf_counter() - t0) / 1e9


print('pinned  -> pitched rows (2D): %.1f GB/s' % rate(lambda: dev2d[:, :g].copy_(pinned, non_blocking=True)))
print('pinned  -> linear       (1D): %.1f GB/s' % rate(lambda: dev1d.copy_(pinned, non_blocking=True)))
pg = torch.from_numpy(pageable)
print('pageable-> pitched rows (2D): %.1f GB/s' % rate(lambda: dev2d[:, :g].copy_(pg)))
print('pageable-> linear       (1D): %.1f GB/s' % rate(lambda: dev1d.copy_(pg)))
print('device linear -> pitched    : %.1f GB/s' % rate(lambda: dev2d[:, :g].copy_(dev1d)))
