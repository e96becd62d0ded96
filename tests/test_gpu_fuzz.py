"""Seeded randomised sweeps over shapes and data regimes (many small cases, one process)."""
import numpy as np
import pytest

import cellcomm_amd as cc
from cellcomm_amd import knn as hipknn
from cellcomm_amd import synth
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


def _data(rng, n, d, kind):
    if kind == 'uniform':
        return rng.random((n, d), dtype=np.float32)
    if kind == 'normal_big':
        return (rng.standard_normal((n, d)) * 1e3 + 5e4).astype(np.float32)      # far from the origin
    if kind == 'tiny':
        return (rng.random((n, d)) * 1e-18).astype(np.float32)
    if kind == 'huge':
        return (rng.random((n, d)) * 1e17).astype(np.float32)                      # squares overflow fp32
    if kind == 'lattice':
        return rng.integers(0, 3, (n, d)).astype(np.float32)                       # masses of exact ties
    if kind == 'clusters':
        c = rng.random((5, d), dtype=np.float32)
        return (c[rng.integers(0, 5, n)] + rng.standard_normal((n, d)).astype(np.float32) * 1e-4).astype(np.float32)
    raise ValueError(kind)


def test_knn_random_shapes_and_regimes(hip, oracle_built):
    rng = np.random.default_rng(2026)
    kinds = ['uniform', 'normal_big', 'tiny', 'huge', 'lattice', 'clusters']
    for case in range(36):
        n = int(rng.integers(2, 2500))
        d = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 100]))
        k = int(min(rng.choice([1, 2, 7, 15, 40, 100]), n - 1))
        include_self = bool(rng.integers(0, 2))
        x = _data(rng, n, d, kinds[case % len(kinds)])
        form = (case // len(kinds)) % 3        # 0 = the library's choice, 1 = norm K-step, 2 = norms through the accumulator
        hipknn.set_collect_norm_form(form)
        try:
            idx, dist = cc.knn_graph(x, k, include_self)
        finally:
            hipknn.set_collect_norm_form(0)
        wi, wd = ko.knn(x, k, include_self)
        assert np.array_equal(idx, wi), (case, n, d, k, kinds[case % len(kinds)], form)
        assert np.array_equal(dist, wd), (case, n, d, k, kinds[case % len(kinds)], form)


def test_knn_large_regimes_sampled_route(hip, oracle_built):
    rng = np.random.default_rng(7)
    for kind in ['normal_big', 'lattice', 'clusters', 'huge']:
        x = _data(rng, 20000, 16, kind)
        hipknn.set_collect_norm_form(2 if kind in ('normal_big', 'clusters') else 1)
        try:
            idx, dist = cc.knn_graph(x, 15)
        finally:
            hipknn.set_collect_norm_form(0)
        for q in rng.choice(20000, 60, replace=False):
            wi, wd = ko.knn(x, 15, q_begin=int(q), n_q=1)
            assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0]), kind


def test_encoder_random_shapes(hip):
    rng = np.random.default_rng(11)
    for case in range(18):
        arch = ['cont', 'cls'][case % 2]
        g = int(rng.integers(1, 700))
        e = int(rng.integers(1, 40))
        n = int(rng.integers(1, 300))
        dtype = ['f32', 'f16x2'][(case // 2) % 2]
        m = synth.synth_csr(n, g, density=float(rng.uniform(0.01, 0.6)), seed=case)
        ws = eo.glorot_weights(arch, e, g, seed=case, randomize_bn=True, bias_scale=0.2)
        enc = cc.HipEncoder(e, g, arch, batch_rows=int(rng.integers(1, 200)), dtype=dtype)
        enc.set_weights(ws)
        out = enc.predict(m)
        truth = eo.forward(arch, ws, m.to_dense())
        rel = float(np.max(np.abs(out - truth) / np.abs(truth)))
        assert rel <= 1e-4, (case, arch, g, e, n, dtype, rel)


def test_gather_random_shapes(hip):
    import ctypes
    from cellcomm_amd import _lib
    rng = np.random.default_rng(5)
    for case in range(20):
        n = int(rng.integers(1, 120))
        g = int(rng.integers(1, 13000))
        m = synth.synth_csr(n, g, density=float(rng.uniform(0.0, 0.3)), seed=case)
        ld = (g + 3) // 4 * 4 + 4 * int(rng.integers(0, 40))
        ids = rng.integers(0, n, int(rng.integers(1, 2 * n + 1))).astype(np.int64)
        out = np.full((len(ids), ld), np.nan, dtype=np.float32)
        _lib.check(hip.cc_gather_dense(m.device_handle(), ids.ctypes.data, 0, len(ids), ld, out.ctypes.data, 0))
        assert np.array_equal(out[:, :g], m.to_dense()[ids]) and np.all(out[:, g:] == 0), case
