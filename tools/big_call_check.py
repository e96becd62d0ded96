#!/usr/bin/env python3
"""GPU-box helper (round 3): the chunked call paths at sizes the test suite does not reach.

  dense fp32   300 000 rows = 37 passes of 8192 -> processed as chunks of 32 + 5 passes (api.hip, encode_csr_dev)
  sparse       300 000 rows = 5 passes of 65 536 rows
  f16        1 100 000 rows = 34 passes of 32 768 -> chunks of 32 + 2

For each: a second call on a row range that straddles the chunk / pass seams must return the same bits, sampled rows
must match the float64 oracle (forward / forward_f16) to 1e-4, and device memory must not grow with the row count
beyond the pass buffers (printed).  G = 30 000, latent 64, 5 % density."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth
from oracle import encoder_oracle as eo

_lib.ensure_init(0)
G, E = 30000, 64


def encode_range(enc, m, lo, hi):
    out = torch.empty((hi - lo, E), dtype=torch.float32, device='cuda')
    enc.predict_to_device(m, out.data_ptr(), lo, hi - lo)
    return out.cpu().numpy()


def check(mode, n, seams):
    m, (indptr, indices, values) = synth.synth_csr_device(n, G, seed=2026 + n % 97)
    enc = cc.HipEncoder(E, G, 'cont', seed=4321, sparse=(mode == 'sparse'), dtype='f16' if mode == 'f16' else 'f32')
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    out = encode_range(enc, m, 0, n)
    dt = time.perf_counter() - t0
    used = (free0 - torch.cuda.mem_get_info()[0]) / 2**30
    assert np.all(np.isfinite(out)) and np.all(out > 0) and np.all(out < 1)
    for lo, hi in seams:
        assert np.array_equal(encode_range(enc, m, lo, hi), out[lo:hi]), (mode, lo, hi)
    rows = np.random.default_rng(1).choice(n, 24, replace=False)
    rows = np.concatenate([rows, [s[0] + (s[1] - s[0]) // 2 for s in seams]])
    ip = indptr.cpu().numpy()
    dense = np.zeros((len(rows), G), dtype=np.float32)
    for o, r in enumerate(rows):
        lo, hi = int(ip[r]), int(ip[r + 1])
        dense[o, indices[lo:hi].cpu().numpy()] = values[lo:hi].cpu().numpy()
    truth = (eo.forward_f16 if mode == 'f16' else eo.forward)('cont', enc.get_weights(), dense)
    rel = float(np.max(np.abs(out[rows] - truth) / np.abs(truth)))
    assert rel <= 1e-4, (mode, rel)
    print(f'{mode:7s} {n:8d} rows: {dt * 1e3:8.1f} ms first call (incl. packing / stream build), +{used:.2f} GiB of device memory, '
          f'max rel err {rel:.2e}, {len(seams)} seam ranges bit-identical', flush=True)
    del enc, m, indptr, indices, values
    torch.cuda.empty_cache()


check('f32', 300000, [(262144 - 300, 262144 + 300), (8192 * 32 - 5, 8192 * 33 + 7), (299000, 300000)])
check('sparse', 300000, [(65536 - 300, 65536 + 300), (262144 - 129, 262144 + 513), (299990, 300000)])
check('f16', 1100000, [(1048576 - 300, 1048576 + 300), (32768 * 32 - 10, 32768 * 33 + 10), (1099000, 1100000)])
print('big calls OK')
