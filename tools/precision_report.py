#!/usr/bin/env python3
"""Max relative error of the encoder against the float64 oracle for each dtype (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cellcomm_amd as cc
from cellcomm_amd import synth
from oracle import encoder_oracle as eo

for arch, g, e, n in [('cont', 30000, 64, 96), ('cont', 2000, 32, 512), ('cls', 30000, 64, 96)]:
    m = synth.synth_csr(n, g, seed=77)
    x = m.to_dense()
    ws = eo.glorot_weights(arch, e, g, seed=4321, randomize_bn=True, bias_scale=0.05)
    truth = eo.forward(arch, ws, x)
    np32 = eo.forward(arch, ws, x, np.float32)
    row = {'numpy_f32': float(np.max(np.abs(np32 - truth) / np.abs(truth)))}
    for dt in ('f32', 'f16x2', 'f16'):
        enc = cc.HipEncoder(e, g, arch, dtype=dt)
        enc.set_weights(ws)
        out = enc.predict(m)
        row[dt] = float(np.max(np.abs(out - truth) / np.abs(truth)))
    for name, kw in (('f32_sparse', dict(sparse=True)), ('f32_low_latency', dict(low_latency=True))):
        enc = cc.HipEncoder(e, g, arch, **kw)
        enc.set_weights(ws)
        out = enc.predict(m)                    # n <= 128 rows take the split-K form with low_latency
        row[name] = float(np.max(np.abs(out - truth) / np.abs(truth)))
    print(arch, g, e, n, {k: f'{v:.2e}' for k, v in row.items()})
