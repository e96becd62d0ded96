#!/usr/bin/env python3
"""Wall time of HipEncoder.predict (the reference seam `encoding_prediction`: encodings into a FRESH pageable numpy array,
reference src/bigan_basic.py:28-30) beside predict_to_device on the same rows (diagnostic).

    python tools/predict_bench.py [cells] [genes] [latent] [dtype] [reps]

The difference of the two is what the host side of the result costs (the device-to-host copy into new pages).  Measured
in round 4: +0.6..1.5 ms for the 25.6 MB of 100k x 64 encodings with the plain synchronous hipMemcpy cc_encode uses — the
pinned-ring + populate form that pays for the kNN results (csrc/knn.hip copy_results) measured the same or slower here
(+1.3..1.4 ms), so the encoder's copy stayed as it was (tools/experiments/README.md)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
e = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dtype = sys.argv[4] if len(sys.argv) > 4 else 'f32'
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
_lib.ensure_init(0)
m, _keep = synth.synth_csr_device(n, g, seed=1234)
enc = cc.HipEncoder(e, g, 'cont', seed=4321, dtype=dtype)
dev = torch.empty((n, e), dtype=torch.float32, device='cuda')
torch.cuda.synchronize()
enc.predict_to_device(m, dev.data_ptr())
ref = dev.cpu().numpy()
out = enc.predict(m)
assert np.array_equal(out, ref), 'predict and predict_to_device disagree'


def wall(fn):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e3 * (time.perf_counter() - t0) / reps


t_dev = wall(lambda: enc.predict_to_device(m, dev.data_ptr()))
t_host = wall(lambda: enc.predict(m))
line = f'n={n} g={g} e={e} {dtype}: predict_to_device {t_dev:.2f} ms, predict {t_host:.2f} ms (+{t_host - t_dev:.2f} for {n * e * 4 / 1e6:.1f} MB of results)'
print(line)
