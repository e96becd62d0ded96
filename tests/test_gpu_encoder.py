"""Encoder forward on the GPU (fp32 MFMA GEMMs) against the float64 oracle.

Tolerance: the north star asks for encodings within 1e-4 relative of the fp32
reference; every comparison below is max |gpu - truth64| / |truth64| <= 1e-4."""
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
def test_golden_fixtures(hip, path):
    z = np.load(path, allow_pickle=False)
    arch, g, e, seed = str(z['arch']), int(z['G']), int(z['E']), int(z['seed'])
    enc = cc.HipEncoder(e, g, arch)
    enc.set_weights(eo.glorot_weights(arch, e, g, seed, randomize_bn=True, bias_scale=0.1))
    out = enc.predict(z['x'])
    assert out.dtype == np.float32 and out.shape == z['out64'].shape
    assert max_rel(out, z['out64']) <= RTOL
    # CSR rows gathered on the GPU give the very same bits as the dense upload
    assert np.array_equal(enc.predict(cc.CellMatrix.from_dense(z['x'])), out)
    # DataFrame / list-of-lists inputs, like keras predict accepts
    assert np.array_equal(enc.predict(z['x'][:3].tolist()), out[:3])


@pytest.mark.parametrize('arch', ['cont', 'cls'])
@pytest.mark.parametrize('g', [1, 4, 5, 11, 19, 20, 33])
def test_degenerate_widths(hip, arch, g):
    # int(G*0.1) == 0 for G < 10 and int(G*0.05) == 0 for G < 20 (reference src/bigan_cont.py:29);
    # the reference's own tests build such models (G in {1, 4, 5, 11}).
    enc = cc.HipEncoder(7, g, arch)
    ws = eo.glorot_weights(arch, 7, g, seed=g, randomize_bn=True, bias_scale=0.2)
    enc.set_weights(ws)
    x = np.random.default_rng(g).integers(0, 6, (9, g)).astype(np.float32)
    assert max_rel(enc.predict(x), eo.forward(arch, ws, x)) <= RTOL


@pytest.mark.parametrize('arch,g,e,n', [('cont', 3000, 64, 700), ('cls', 3000, 10, 700), ('cont', 777, 3, 1291)])
def test_seeded_mid_size_vs_oracle(hip, arch, g, e, n):
    m = synth.synth_csr(n, g, seed=g + n)
    ws = eo.glorot_weights(arch, e, g, seed=9, randomize_bn=True, bias_scale=0.05)
    enc = cc.HipEncoder(e, g, arch, batch_rows=512)     # forces several passes + a ragged tail
    enc.set_weights(ws)
    out = enc.predict(m)
    assert max_rel(out, eo.forward(arch, ws, m.to_dense())) <= RTOL
    ids = np.random.default_rng(1).choice(n, 129, replace=False)
    assert np.array_equal(enc.predict(m.take(ids)), out[ids])       # minibatch = same rows, same bits


def test_full_width_rows_vs_oracle(hip):
    # BASELINE config 3 layer shapes (G=30000 -> 3000 -> 1500 -> 150 -> 150 -> 64) on 48 cells
    g, e, n = 30000, 64, 48
    m = synth.synth_csr(n, g, seed=77)
    enc = cc.HipEncoder(e, g, 'cont', seed=4321)
    ws = enc.get_weights()
    out = enc.predict(m)
    assert max_rel(out, eo.forward('cont', ws, m.to_dense())) <= RTOL


def test_fresh_model_uses_keras_defaults(hip):
    enc = cc.HipEncoder(12, 100, 'cont', seed=0)
    ws = enc.get_weights()
    assert [w.shape for w in ws] == [tuple(s) for s in eo.weight_shapes('cont', 12, 100)]
    for i, shp in enumerate(eo.weight_shapes('cont', 12, 100)):
        if len(shp) == 2:
            assert np.abs(ws[i]).max() <= np.sqrt(6.0 / sum(shp)) and ws[i].std() > 0
    assert np.all(ws[4] == 1) and np.all(ws[5] == 0) and np.all(ws[6] == 0) and np.all(ws[7] == 1)
    assert all(np.all(ws[i] == 0) for i in (1, 3, 9, 11, 13))
    assert enc.input_shape == (None, 100) and enc.output_shape == (None, 12)
    assert enc.layers[-1].activation == 'sigmoid' and cc.HipEncoder(12, 100, 'cls').layers[-1].activation == 'softmax'
    assert len(enc.layers[1].get_weights()) == 2 and len(enc.layers[6].get_weights()) == 4
    enc.summary(print_fn=lambda s: None)


def test_set_weights_changes_output_and_validates(hip):
    enc = cc.HipEncoder(5, 64, 'cls', seed=1)
    x = np.random.default_rng(0).integers(0, 5, (4, 64)).astype(np.float32)
    before = enc.predict(x)
    assert np.allclose(before.sum(-1), 1.0, atol=1e-6)
    ws = enc.get_weights()
    ws[-1] = ws[-1] + np.arange(5, dtype=np.float32)
    enc.set_weights(ws)
    assert not np.array_equal(enc.predict(x), before)
    with pytest.raises(ValueError):
        enc.set_weights(ws[:-1])
    with pytest.raises(ValueError):
        enc.set_weights([w.T for w in ws])
    with pytest.raises(ValueError):
        enc.predict(np.zeros((2, 63), dtype=np.float32))
    assert enc.predict(np.zeros((0, 64), dtype=np.float32)).shape == (0, 5)


def test_row_results_do_not_depend_on_batching(hip):
    # size-independent property used at full scale: a cell's encoding depends only on its own row
    g, n = 4096, 1500
    m = synth.synth_csr(n, g, seed=3)
    a = cc.HipEncoder(16, g, 'cont', seed=5, batch_rows=8192)
    b = cc.HipEncoder(16, g, 'cont', seed=5, batch_rows=333)
    out = a.predict(m)
    assert np.array_equal(b.predict(m), out)
    perm = np.random.default_rng(2).permutation(n)
    assert np.array_equal(a.predict(m.take(perm)), out[perm])


@pytest.mark.parametrize('n', [4097, 12288, 13001, 20000])
def test_dense_host_rows_in_uneven_passes(hip, n):
    """cc_encode_dense copies pass i + 1 under the kernels of pass i (two slots, 4096-row passes or the encoder's own
    pass size when that is smaller).  Whatever the cut, a row's encoding is the encoding of the same row from the CSR
    matrix, bit for bit."""
    g, e = 1500, 16
    m = synth.synth_csr(n, g, seed=n)
    enc = cc.HipEncoder(e, g, 'cont', seed=9)
    want = enc.predict(m)
    dense = m.to_dense()
    assert np.array_equal(enc.predict(dense), want)
    few = cc.HipEncoder(e, g, 'cont', seed=9, batch_rows=1024)        # 1024-row passes, a ragged last one
    assert np.array_equal(few.predict(dense), want)
    rows = np.random.default_rng(n).choice(n, 40, replace=False)
    assert max_rel(want[rows], eo.forward('cont', enc.get_weights(), dense[rows])) <= RTOL
