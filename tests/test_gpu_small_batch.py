"""Opt-in low-latency form of small calls (<= 128 rows: keras predict batches of 32, reference
src/bigan_basic.py:30; minibatches of 128, src/__main__.py:44): layers 1-2 as split-K over ~1024 workgroups with a
fixed-order reduction (csrc/gemm_small.hip).  Same 1e-4 bar against the float64 oracle, reproducible call to call;
larger calls and encoders without the option keep the tiled kernel bit for bit."""
import numpy as np
import pytest

import cellcomm_amd as cc
from cellcomm_amd import synth
from oracle import encoder_oracle as eo

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def max_rel(got, truth):
    return float(np.max(np.abs(got.astype(np.float64) - truth) / np.abs(truth)))


@pytest.mark.parametrize('arch,g,e', [('cont', 30000, 64), ('cls', 30000, 10), ('cont', 2500, 16), ('cont', 6000, 3),
                                      ('cont', 12000, 32), ('cont', 11000, 130),   # fused tail, 1 / 2 output tiles
                                      ('cont', 8000, 300)])                         # ... and 3 (encoding_size > 256)
def test_low_latency_small_calls_match_the_oracle(hip, arch, g, e):
    m = synth.synth_csr(200, g, seed=g + e)
    ws = eo.glorot_weights(arch, e, g, seed=5, randomize_bn=True, bias_scale=0.05)
    fast = cc.HipEncoder(e, g, arch, low_latency=True)
    fast.set_weights(ws)
    base = cc.HipEncoder(e, g, arch)
    base.set_weights(ws)
    dense = m.to_dense()
    truth = eo.forward(arch, ws, dense)
    full = base.predict(m)
    for n in (1, 7, 32, 33, 96, 128):
        out = fast.predict(dense[:n])                       # dense host rows, as keras predict gets them
        assert out.shape == (n, e) and max_rel(out, truth[:n]) <= RTOL
        assert np.array_equal(fast.predict(dense[:n]), out)                 # fixed reduction order: same bits every call
        assert np.array_equal(fast.predict(m.take(np.arange(n))), out)      # CSR rows take the same kernels
        assert max_rel(out, full[:n].astype(np.float64)) <= 2e-5            # vs the tiled kernel: fp32 rounding only
    # above 128 rows the option changes nothing, and an encoder without it is batch independent bit for bit
    assert np.array_equal(fast.predict(dense[:129]), full[:129])
    assert np.array_equal(base.predict(dense[:32]), full[:32])
    assert np.array_equal(fast.predict(m), full)


def test_low_latency_minibatch_of_sampled_rows(hip):
    g, e = 30000, 64
    m = synth.synth_csr(3000, g, seed=8)
    enc = cc.HipEncoder(e, g, 'cont', seed=4321, low_latency=True)
    batch = m.sample(128, random_state=0)                   # CellTraining.sample_cell_data
    out = enc.predict(batch)
    truth = eo.forward('cont', enc.get_weights(), batch.to_dense())
    assert max_rel(out, truth) <= RTOL
