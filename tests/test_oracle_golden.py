"""The oracle against the reference's own known answers and against itself
(NumPy vs C restatement vs sklearn), plus: committed fixtures == oracle output."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_npz
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko
from oracle import loader_oracle as lo
from oracle import schema_oracle as so


def test_load_matrix_known_answer(known):
    df = lo.load_matrix(os.path.join(GOLDEN, 'example_matrix.mtx'))
    assert list(df.shape) == known['load_matrix']['shape']
    assert np.array_equal(df.to_numpy(), np.array(known['load_matrix']['matrix']))


def test_sample_rows_known_answer(known):
    df = lo.load_matrix(os.path.join(GOLDEN, 'example_matrix.mtx'))
    ka = known['sample_cell_data']
    sampled = lo.sample_cell_data(df, ka['batch_size'], ka['seed'])
    expect = np.array(known['load_matrix']['matrix'])[ka['rows']]
    assert np.array_equal(sampled.to_numpy(), expect)


def test_encits_document_known_answer(known):
    ka = known['encits_document']
    doc = so.encits_document('run', 2009, np.array(ka['encodings']), ka['ns'])
    for key in ('cids', 'ns', 'xs', 'ys', 'zs', 'ds'):
        assert doc[key] == ka[key], key


def test_schema_messages(known):
    msg = known['messages']
    with pytest.raises(AssertionError) as e:
        so.encits_document('r', 1, np.array([[], [], [], []]), ['a'] * 5)
    assert str(e.value) == msg['length']
    with pytest.raises(AssertionError) as e:
        so.encits_document('r', 1, np.array([[1, 2]] * 5), ['a'] * 5)
    assert str(e.value) == msg['xyz']
    with pytest.raises(ValueError) as e:
        so.legacy_encs_document('1', np.zeros((4, 3)), ['a'] * 5)
    assert str(e.value) == msg['legacy_length']


def test_onehot_known_answer(known):
    ka = known['trainings_encoding_prediction_onehot']
    assert np.array_equal(eo.trainings_encoding_prediction_cls(ka['prediction'], 2), np.array(ka['onehot']))


@pytest.mark.parametrize('arch,g,e', [('cont', 100, 12), ('cls', 100, 12), ('cont', 5, 8), ('cont', 11, 7), ('cont', 1, 20)])
def test_encoder_shapes_and_ranges(arch, g, e):
    ws = eo.glorot_weights(arch, e, g, seed=3)
    assert [w.shape for w in ws] == [tuple(s) for s in eo.weight_shapes(arch, e, g)]
    x = np.random.default_rng(0).integers(0, 4, (6, g)).astype(np.float32)
    out = eo.forward(arch, ws, x)
    assert out.shape == (6, e)
    assert np.all(out > 0) and np.all(out < 1)
    if arch == 'cls':
        assert np.allclose(out.sum(-1), 1.0)


def test_degenerate_widths_give_constant_rows():
    # G < 20 => int(G*0.05) == 0 => the third Dense sees a zero-width input
    # (reference src/bigan_cont.py:29,38): every cell maps to the same encoding.
    ws = eo.glorot_weights('cont', 8, 5, seed=1, bias_scale=0.3)
    out = eo.forward('cont', ws, np.arange(15, dtype=np.float32).reshape(3, 5))
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[1], out[2])


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'encoder_*.npz'))))
def test_encoder_fixtures_match_oracle(path, oracle_built):
    z = np.load(path, allow_pickle=False)
    arch, g, e, seed = str(z['arch']), int(z['G']), int(z['E']), int(z['seed'])
    ws = eo.glorot_weights(arch, e, g, seed, randomize_bn=True, bias_scale=0.1)
    wsum = float(sum(np.abs(w.astype(np.float64)).sum() for w in ws))
    assert wsum == float(z['wsum']), 'NumPy RNG stream differs from the one that made the fixture'
    out64 = eo.forward(arch, ws, z['x'], np.float64)
    assert np.allclose(out64, z['out64'], rtol=1e-12, atol=0)
    # three independent float32 evaluations agree with the float64 truth to 1e-5 relative
    for approx in (z['out32'], eo.predict_like_keras(arch, ws, z['x']), ko.encoder_forward_c(arch, ws, z['x'], e)):
        assert np.max(np.abs(approx - out64) / np.abs(out64)) < 1e-5


def torch_modules_forward(arch, ws, x, dtype):
    """The same graphs out of torch.nn MODULES (torch's own Linear / BatchNorm1d(eps=1e-3).eval() / Sigmoid / Softmax):
    a fourth statement of the Keras op definitions by the only other framework in this image that implements
    BatchNormalization inference and softmax itself.  Layer order: reference src/bigan_cont.py:31-41,
    src/bigan_classify.py:29-40.  (It does not pin the oracle to TensorFlow — nothing here can.)"""
    import torch
    from torch import nn

    def dense(k, b):
        lin = nn.Linear(k.shape[0], k.shape[1]).to(dtype)
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(np.ascontiguousarray(k.T)))   # Keras kernels are [in x out]
            lin.bias.copy_(torch.from_numpy(b))
        return lin

    xt = torch.from_numpy(np.asarray(x, dtype=np.float32)).to(dtype)        # predict() casts the input to float32 first
    sig = nn.Sigmoid()
    with torch.no_grad():
        if arch == 'cont':
            k1, b1, k2, b2, gamma, beta, mean, var, k3, b3, k4, b4, k5, b5 = ws
            bn = nn.BatchNorm1d(len(gamma), eps=1e-3).to(dtype)
            bn.weight.copy_(torch.from_numpy(gamma)); bn.bias.copy_(torch.from_numpy(beta))
            bn.running_mean.copy_(torch.from_numpy(mean)); bn.running_var.copy_(torch.from_numpy(var))
            bn.eval()
            h = nn.Dropout(0.15).eval()(sig(dense(k1, b1)(xt)))
            h = nn.Dropout(0.1).eval()(sig(dense(k2, b2)(torch.cat([h, xt], dim=-1))))
            h = bn(h)
            h = sig(dense(k3, b3)(h))
            h = sig(dense(k4, b4)(h))
            return sig(dense(k5, b5)(h)).numpy()
        k1, b1, k2, b2, k3, b3, k4, b4 = ws
        p = nn.Dropout(0.1).eval()(sig(dense(k1, b1)(xt)))
        h = nn.Dropout(0.15).eval()(sig(dense(k2, b2)(p)))
        h = sig(dense(k3, b3)(torch.cat([h, p], dim=-1)))
        return nn.Softmax(dim=-1)(dense(k4, b4)(h)).numpy()


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'encoder_*.npz'))))
def test_encoder_fixtures_match_torch_modules(path):
    import torch
    z = np.load(path, allow_pickle=False)
    arch, g, e, seed = str(z['arch']), int(z['G']), int(z['E']), int(z['seed'])
    ws = eo.glorot_weights(arch, e, g, seed, randomize_bn=True, bias_scale=0.1)
    if any(w.size == 0 for w in ws):
        pytest.skip('degenerate widths (int(G*0.05) == 0): torch.nn.Linear with zero input features is covered by the oracle only')
    out64 = torch_modules_forward(arch, ws, z['x'], torch.float64)
    assert np.max(np.abs(out64 - z['out64']) / np.abs(z['out64'])) < 1e-12
    out32 = torch_modules_forward(arch, ws, z['x'], torch.float32)
    assert np.max(np.abs(out32 - z['out64']) / np.abs(z['out64'])) < 1e-5


@pytest.mark.parametrize('name', ['knn_N257_E3.npz', 'knn_N2048_E32.npz', 'knn_N2048_E64.npz'])
def test_knn_fixtures_match_oracle(name, oracle_built):
    z = load_npz(name)
    for k in (15, 50):
        idx, dist = ko.knn(z['x'], k)
        assert np.array_equal(idx, z[f'idx{k}'])
        assert np.array_equal(dist, z[f'dist{k}'])


def test_knn_c_equals_numpy_statement(oracle_built):
    rng = np.random.default_rng(5)
    x = rng.random((400, 9), dtype=np.float32)
    x[100:110] = x[7]
    for include_self in (False, True):
        a = ko.knn(x, 12, include_self)
        b = ko.knn_numpy(x, 12, include_self)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_knn_against_sklearn_on_tie_free_data(oracle_built):
    from sklearn.neighbors import NearestNeighbors
    x = np.random.default_rng(11).random((1500, 16), dtype=np.float32)
    idx, dist = ko.knn(x, 15)
    nn = NearestNeighbors(n_neighbors=16, algorithm='brute', metric='sqeuclidean').fit(x.astype(np.float64))
    sk_d, sk_i = nn.kneighbors(x.astype(np.float64))
    assert np.array_equal(sk_i[:, 0], np.arange(len(x)))          # self first, then the same neighbours
    assert np.array_equal(idx, sk_i[:, 1:].astype(np.int32))
    assert np.allclose(dist, sk_d[:, 1:], rtol=1e-5)


def test_knn_ties_break_by_index_and_self_by_index(oracle_built):
    x = np.zeros((6, 3), dtype=np.float32)
    x[4] = 1.0
    idx, dist = ko.knn(x, 3)
    assert idx[0].tolist() == [1, 2, 3] and idx[2].tolist() == [0, 1, 3]
    assert idx[4].tolist() == [0, 1, 2] and np.all(dist[4] == 3.0)
    idx_s, _ = ko.knn(x, 3, include_self=True)
    assert idx_s[2].tolist() == [0, 1, 2]
    with pytest.raises(ValueError):
        ko.knn(x, 6)
