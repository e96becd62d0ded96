"""Sparse-aware first layer (csrc/spmm.hip): CSR rows x fp32 weights over the non-zeros only.
Same bar as the dense path: <= 1e-4 relative against the float64 oracle; plus agreement with the dense path,
row-range handling, stream edge cases (empty rows, dense rows, ragged last block, rows with > 128 non-zeros in
one 64-column stage) and the gather + dense fallback for sampled rows."""
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
def test_sparse_on_golden_fixtures(hip, path):
    z = np.load(path, allow_pickle=False)
    arch, g, e, seed = str(z['arch']), int(z['G']), int(z['E']), int(z['seed'])
    enc = cc.HipEncoder(e, g, arch, sparse=True)
    enc.set_weights(eo.glorot_weights(arch, e, g, seed, randomize_bn=True, bias_scale=0.1))
    out = enc.predict(cc.CellMatrix.from_dense(z['x']))
    assert max_rel(out, z['out64']) <= RTOL


@pytest.mark.parametrize('arch,g,e,n', [('cont', 3000, 64, 700), ('cls', 3000, 10, 700), ('cont', 777, 3, 1291),
                                        ('cont', 64, 8, 300), ('cont', 20, 5, 33), ('cls', 5, 4, 40),
                                        ('cont', 5000, 32, 256), ('cont', 2563, 16, 257), ('cls', 2563, 16, 1023)])
def test_sparse_seeded_vs_oracle_and_dense_path(hip, arch, g, e, n):
    m = synth.synth_csr(n, g, seed=g + n)
    ws = eo.glorot_weights(arch, e, g, seed=9, randomize_bn=True, bias_scale=0.05)
    sp = cc.HipEncoder(e, g, arch, sparse=True)
    sp.set_weights(ws)
    out = sp.predict(m)
    truth = eo.forward(arch, ws, m.to_dense())
    assert max_rel(out, truth) <= RTOL
    dn = cc.HipEncoder(e, g, arch)
    dn.set_weights(ws)
    assert max_rel(out, dn.predict(m).astype(np.float64)) <= 2e-5          # two fp32 evaluations of the same sums
    assert np.array_equal(sp.predict(m), out)                               # deterministic, bit for bit
    # sampled rows (minibatch) and dense host input keep the gather + dense path: same bits as the dense encoder
    ids = np.random.default_rng(1).choice(n, min(129, n), replace=False)
    assert np.array_equal(sp.predict(m.take(ids)), dn.predict(m.take(ids)))
    assert np.array_equal(sp.predict(m.to_dense()[:50]), dn.predict(m.to_dense()[:50]))


def test_sparse_row_ranges_do_not_depend_on_the_call(hip):
    """A row's encoding is the same whichever contiguous range of the matrix it is encoded with (blocks of 256 rows
    are cut at multiples of 256 of the MATRIX, ranges may start and end anywhere)."""
    import torch
    g, e, n = 1500, 16, 2000
    m = synth.synth_csr(n, g, seed=5)
    enc = cc.HipEncoder(e, g, 'cont', seed=3, sparse=True)
    full = enc.predict(m)
    for lo, cnt in [(0, 1), (255, 2), (256, 256), (100, 1000), (1999, 1), (513, 1487), (7, 0)]:
        out = torch.empty((max(cnt, 1), e), dtype=torch.float32, device='cuda')
        enc.predict_to_device(m, out.data_ptr(), lo, cnt)
        assert np.array_equal(out.cpu().numpy()[:cnt], full[lo:lo + cnt]), (lo, cnt)


def test_sparse_stream_edge_cases(hip):
    g, e = 700, 8
    rng = np.random.default_rng(4)
    n = 600
    x = ((rng.random((n, g)) < 0.05) * rng.integers(1, 40, (n, g))).astype(np.float32)
    x[0] = 0                                   # empty first row
    x[17] = 0
    x[255] = rng.integers(1, 9, g)             # fully dense rows: 64 entries of one row in every stage
    x[256] = rng.integers(1, 9, g)
    x[300:316, 64:128] = rng.integers(1, 5, (16, 64))   # a 16-row group with 1024 entries in ONE stage (> 128: window refetch)
    x[590:] = 0                                # empty tail rows
    x[599, g - 1] = 3.0                        # last column of the last (ragged) stage
    x[5, 3] = 0.125                            # values need not be integers
    x[6, 4] = -2.5
    ws = eo.glorot_weights('cont', e, g, seed=1, randomize_bn=True, bias_scale=0.1)
    enc = cc.HipEncoder(e, g, 'cont', sparse=True)
    enc.set_weights(ws)
    m = cc.CellMatrix.from_dense(x)
    out = enc.predict(m)
    assert max_rel(out, eo.forward('cont', ws, x)) <= RTOL
    empty = cc.CellMatrix.from_dense(np.zeros((300, g), dtype=np.float32))      # a matrix without any non-zero
    assert max_rel(enc.predict(empty), eo.forward('cont', ws, np.zeros((300, g)))) <= RTOL
    assert enc.predict(m.take(np.arange(0))).shape == (0, e)


def test_sparse_full_width_rows_and_two_matrices(hip):
    """G = 30000 (the bench width: 469 stages + a ragged one, 12 / 6 column panels), two matrices alternating on one
    encoder (the stream belongs to the matrix handle)."""
    g, e = 30000, 64
    enc = cc.HipEncoder(e, g, 'cont', seed=4321, sparse=True)
    ws = enc.get_weights()
    a = synth.synth_csr(300, g, seed=77)
    b = synth.synth_csr(47, g, seed=78)
    oa, ob = enc.predict(a), enc.predict(b)
    assert max_rel(oa[:40], eo.forward('cont', ws, a.to_dense()[:40])) <= RTOL
    assert max_rel(ob, eo.forward('cont', ws, b.to_dense())) <= RTOL
    assert np.array_equal(enc.predict(a), oa) and np.array_equal(enc.predict(b), ob)


def test_sparse_needs_fp32_dtype(hip):
    with pytest.raises(ValueError):
        cc.HipEncoder(8, 100, 'cont', dtype='f16x2', sparse=True)
