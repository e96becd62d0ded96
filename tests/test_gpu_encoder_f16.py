"""Layers 1-2 on the fp16 matrix cores.

dtype 'f16x2' (hi+lo split operands) must meet the SAME tolerance as the fp32 path: <= 1e-4 relative
against the float64 oracle.  dtype 'f16' (operands rounded to fp16 once, BASELINE config 5) is
checked <= 1e-4 relative against the float64 evaluation of exactly those rounded operands
(oracle.encoder_oracle.forward_f16); its distance to the unrounded truth is reported, not asserted."""
import glob
import os

import numpy as np
import pytest

import cellcomm_amd as cc
from conftest import GOLDEN
from cellcomm_amd import synth
from oracle import encoder_oracle as eo

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def max_rel(got, truth):
    return float(np.max(np.abs(got.astype(np.float64) - truth) / np.abs(truth)))


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'encoder_*.npz'))))
def test_f16x2_on_golden_fixtures(hip, path):
    z = np.load(path, allow_pickle=False)
    arch, g, e, seed = str(z['arch']), int(z['G']), int(z['E']), int(z['seed'])
    enc = cc.HipEncoder(e, g, arch, dtype='f16x2')
    enc.set_weights(eo.glorot_weights(arch, e, g, seed, randomize_bn=True, bias_scale=0.1))
    out = enc.predict(cc.CellMatrix.from_dense(z['x']))
    assert enc.fp32_fallback_passes() == 0                 # integer counts are exact in fp16
    assert max_rel(out, z['out64']) <= RTOL
    assert np.array_equal(enc.predict(z['x']), out)        # dense host input takes the same kernels


@pytest.mark.parametrize('arch,g,e,n', [('cont', 3000, 64, 700), ('cls', 3000, 10, 700), ('cont', 777, 3, 1291),
                                        ('cont', 19, 5, 33), ('cont', 64, 8, 300)])
def test_f16x2_seeded_vs_oracle(hip, arch, g, e, n):
    m = synth.synth_csr(n, g, seed=g + n)
    ws = eo.glorot_weights(arch, e, g, seed=9, randomize_bn=True, bias_scale=0.05)
    enc = cc.HipEncoder(e, g, arch, batch_rows=512, dtype='f16x2')
    enc.set_weights(ws)
    out = enc.predict(m)
    assert max_rel(out, eo.forward(arch, ws, m.to_dense())) <= RTOL
    ids = np.random.default_rng(1).choice(n, min(129, n), replace=False)
    assert np.array_equal(enc.predict(m.take(ids)), out[ids])


def test_f16x2_full_width_rows(hip):
    g, e, n = 30000, 64, 48
    m = synth.synth_csr(n, g, seed=77)
    enc = cc.HipEncoder(e, g, 'cont', seed=4321, dtype='f16x2')
    out = enc.predict(m)
    truth = eo.forward('cont', enc.get_weights(), m.to_dense())
    assert max_rel(out, truth) <= RTOL
    ref32 = cc.HipEncoder(e, g, 'cont', seed=4321).predict(m)
    assert max_rel(ref32, truth) <= RTOL


def test_f16x2_falls_back_to_fp32_on_inexact_input(hip):
    g, n = 500, 200
    rng = np.random.default_rng(3)
    x = (rng.random((n, g)) < 0.1) * rng.integers(1, 9, (n, g))
    x = x.astype(np.float32)
    x[5, 7] = 0.1            # not an fp16 number
    x[150, 3] = 4097.0       # beyond fp16's integer range
    ws = eo.glorot_weights('cont', 6, g, seed=2, randomize_bn=True, bias_scale=0.1)
    enc = cc.HipEncoder(6, g, 'cont', batch_rows=64, dtype='f16x2')
    enc.set_weights(ws)
    out = enc.predict(cc.CellMatrix.from_dense(x))
    assert enc.fp32_fallback_passes() == 2                 # the passes holding rows 5 and 150
    assert max_rel(out, eo.forward('cont', ws, x)) <= RTOL
    assert max_rel(enc.predict(x), eo.forward('cont', ws, x)) <= RTOL
    assert enc.fp32_fallback_passes() == 2


@pytest.mark.parametrize('arch,g,e,n', [('cont', 2000, 32, 300), ('cls', 2000, 12, 300), ('cont', 30000, 64, 40)])
def test_f16_operands_mode(hip, arch, g, e, n):
    m = synth.synth_csr(n, g, seed=g)
    ws = eo.glorot_weights(arch, e, g, seed=11, randomize_bn=True, bias_scale=0.05)
    enc = cc.HipEncoder(e, g, arch, dtype='f16')
    enc.set_weights(ws)
    out = enc.predict(m)
    x = m.to_dense()
    assert max_rel(out, eo.forward_f16(arch, ws, x)) <= RTOL          # what the mode is defined to compute
    drift = max_rel(out, eo.forward(arch, ws, x))                     # vs unrounded operands: informational
    print(f'f16 operands vs float64 truth: {drift:.2e} relative')
    assert drift < 5e-2
