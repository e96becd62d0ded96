"""Generator forward (generate_cells, reference src/bigan_basic.py:39-44) on the GPU against the
float64 oracle.  Outputs pass through ReLU, so the bound is |err| <= 1e-4·|truth| + 1e-6."""
import numpy as np
import pytest

import cellcomm_amd as cc
from oracle import encoder_oracle as eo

pytestmark = pytest.mark.gpu


def check(got, truth):
    err = np.abs(got.astype(np.float64) - truth)
    assert np.all(err <= 1e-4 * np.abs(truth) + 1e-6), float(np.max(err - 1e-4 * np.abs(truth)))


@pytest.mark.parametrize('arch,g,e,n', [('cont', 100, 13, 40), ('cls', 1000, 15, 64), ('cont', 3000, 8, 300),
                                        ('cls', 30, 4, 7), ('cont', 4, 3, 5), ('cont', 11, 7, 9)])
def test_generator_vs_oracle(hip, arch, g, e, n):
    ws = eo.generator_weights(arch, e, g, seed=g + e)
    gen = cc.HipGenerator(e, g, arch)
    assert [w.shape for w in gen.get_weights()] == [tuple(s) for s in eo.generator_weight_shapes(arch, e, g)]
    gen.set_weights(ws)
    rng = np.random.default_rng(n)
    enc_in = rng.random((n, e), dtype=np.float32)
    rand_in = rng.random((n, e), dtype=np.float32)
    out = gen.predict((enc_in, rand_in))
    assert out.shape == (n, g) and out.dtype == np.float32 and np.all(out >= 0)
    check(out, eo.generator_forward(arch, ws, enc_in, rand_in))
    assert gen.layers[-1].activation == 'relu' and gen.output_shape == (None, g)      # reference test/bigans_cc_test.py:24-25


def test_generate_cells_through_the_bigan(hip):
    g, e = 600, 5
    net = cc.ContinuousCellBiGan(e, g, generator_factory=cc.build_cont_generator)
    ws = eo.generator_weights('cont', e, g, seed=1, bias_scale=2.0)          # large biases -> counts above 0.5
    net._generator.set_weights(ws)
    enc_in = np.eye(e, dtype=np.float32)[[0, 3, 4]]
    noise = net.random_uniform_vector(3)
    cells = net.generate_cells(enc_in, noise)
    truth = eo.generator_forward('cont', ws, enc_in, noise)
    assert cells.shape == (3, g)
    safe = np.abs(truth - np.floor(truth) - 0.5) > 1e-3                     # away from rounding boundaries
    assert np.array_equal(cells[safe], np.round(truth)[safe]) and cells.max() >= 1
    assert net.generate_cells(enc_in).shape == (3, g)                        # noise drawn internally
    with pytest.raises(ValueError):
        net._generator.predict((enc_in[:, :3], noise))
    with pytest.raises(NotImplementedError):
        cc.ContinuousCellBiGan(e, g).generate_cells(enc_in, noise)           # default: generator not built
