#!/usr/bin/env python3
"""Race screen of the fp16 ping-pong K loops (csrc/gemm16.hip): for a time budget, over several shapes, encode a long row
range (whole rounds of 256 x 256 tiles: the ping-pong loops, counted waits on LDS-DMA) repeatedly and compare every
output bit with the same rows encoded 2048 at a time (half-height tiles: the older loop) — same MFMAs in the same order,
so any difference is a misplaced wait or barrier.  usage: pp_screen.py [seconds] [seed]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
_lib.ensure_init(0)
t_end = time.time() + budget
shapes = [(65536, 30000), (40000, 33000), (70000, 20000), (33000, 12000), (66000, 8192), (50000, 31744)]
runs = compared = 0
while time.time() < t_end:
    n, g = shapes[runs % len(shapes)]
    n = int(n - rng.integers(0, 300))                       # ragged last tile
    e = int(rng.choice([8, 64, 100]))
    dtype = ['f16', 'f16x2'][runs % 2]
    arch = 'cont' if runs % 5 else 'cls'
    m, _keep = synth.synth_csr_device(n, g, density=float(rng.choice([0.02, 0.05])), seed=int(rng.integers(1 << 30)))
    enc = cc.HipEncoder(e, g, arch, seed=int(rng.integers(1 << 30)), dtype=dtype)
    out = torch.empty((n, e), dtype=torch.float32, device='cuda')
    enc.predict_to_device(m, out.data_ptr(), 0, n)
    whole = out.clone()
    for rep in range(3):                                    # the long call again: same bits
        enc.predict_to_device(m, out.data_ptr(), 0, n)
        assert torch.equal(out, whole), f'{dtype} {n}x{g} e={e}: run {rep} of the long call differs from the first'
    part = torch.empty((2048, e), dtype=torch.float32, device='cuda')
    for lo in range(0, n, 2048):
        cnt = min(2048, n - lo)
        enc.predict_to_device(m, part.data_ptr(), lo, cnt)
        if not torch.equal(part[:cnt], whole[lo:lo + cnt]):
            bad = (part[:cnt] != whole[lo:lo + cnt]).nonzero()[0].tolist()
            raise SystemExit(f'MISMATCH {dtype} {arch} {n}x{g} e={e}: rows {lo}+ first at {bad}')
        compared += cnt
    assert enc.fp32_fallback_passes() == 0
    runs += 1
    print(f'{runs}: {dtype} {arch} {n} x {g}, e = {e}: identical', flush=True)
    del m, _keep, enc, out, whole, part
    gc.collect()
    torch.cuda.empty_cache()           # the library allocates with hipMalloc beside torch's caching allocator
print(f'pp_screen: {runs} shapes, {compared} rows compared bit for bit, no difference (seed {seed})')
