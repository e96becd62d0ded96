#!/usr/bin/env python3
"""Randomised soak of the GPU path against the oracle for a time budget (seconds).  Not part of the test suite:
a longer, wider version of tests/test_gpu_fuzz.py to run when GPU time is spare.  Prints a line every few cases;
exits non-zero at the first mismatch with the offending case."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import knn as hipknn, synth
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print('soak seed', seed, 'budget', budget, flush=True)
t_end = time.time() + budget
kinds = ['uniform', 'normal_big', 'tiny', 'huge', 'lattice', 'clusters', 'tight_sorted', 'dup_heavy']


def data(n, d, kind):
    if kind == 'uniform': return rng.random((n, d), dtype=np.float32)
    if kind == 'normal_big': return (rng.standard_normal((n, d)) * 1e3 + 5e4).astype(np.float32)
    if kind == 'tiny': return (rng.random((n, d)) * 1e-18).astype(np.float32)
    if kind == 'huge': return (rng.random((n, d)) * 1e17).astype(np.float32)
    if kind == 'lattice': return rng.integers(0, 3, (n, d)).astype(np.float32)
    if kind == 'clusters':
        c = rng.random((5, d), dtype=np.float32)
        return (c[rng.integers(0, 5, n)] + rng.standard_normal((n, d)).astype(np.float32) * 1e-4).astype(np.float32)
    if kind == 'tight_sorted':
        nc = max(1, n // int(rng.integers(8, 400)))
        c = rng.random((nc, d), dtype=np.float32) * 10
        lab = np.sort(rng.integers(0, nc, n))
        return (c[lab] + rng.standard_normal((n, d)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-4, -0.5))).astype(np.float32)
    if kind == 'dup_heavy':
        sites = rng.random((int(rng.integers(2, 50)), d), dtype=np.float32)
        return sites[rng.integers(0, len(sites), n)]
    raise ValueError(kind)


cases = 0
while time.time() < t_end:
    what = rng.choice(['knn_small', 'knn_big', 'encoder', 'shards', 'sparse', 'dense_host'], p=[0.3, 0.15, 0.2, 0.1, 0.2, 0.05])
    if what == 'knn_small':
        n = int(rng.integers(2, 4000)); d = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 100, 132]))
        k = int(min(rng.choice([1, 2, 7, 15, 40, 100, 128]), n - 1)); inc = bool(rng.integers(0, 2)); kind = str(rng.choice(kinds))
        x = data(n, d, kind)
        form = int(rng.integers(0, 3))                 # collect tile: the library's choice / norm K-step / norms through the accumulator
        hipknn.set_collect_norm_form(form)
        idx, dist = cc.knn_graph(x, k, inc)
        hipknn.set_collect_norm_form(0)
        wi, wd = ko.knn(x, k, inc)
        ok = np.array_equal(idx, wi) and np.array_equal(dist, wd)
        desc = (what, n, d, k, inc, kind, form)
    elif what == 'knn_big':
        n = int(rng.integers(8000, 60000)); d = int(rng.choice([3, 16, 32, 64])); k = int(rng.choice([1, 15, 50])); kind = str(rng.choice(kinds))
        x = data(n, d, kind)
        q0 = int(rng.integers(0, n // 2)); nq = int(rng.integers(1, n - q0))
        t = torch.from_numpy(x).cuda(); torch.cuda.synchronize()
        form = int(rng.integers(0, 3))
        hipknn.set_collect_norm_form(form)
        idx, dist = hipknn.knn_graph_device(t.data_ptr(), n, d, k, q_begin=q0, n_q=nq)
        hipknn.set_collect_norm_form(0)
        ok = True
        for ql in rng.choice(nq, min(40, nq), replace=False):
            wi, wd = ko.knn(x, k, q_begin=q0 + int(ql), n_q=1)
            ok = ok and np.array_equal(idx[ql], wi[0]) and np.array_equal(dist[ql], wd[0])
        desc = (what, n, d, k, kind, q0, nq, form, hipknn.last_fallback_count(), hipknn.last_full_scan_count())
    elif what == 'encoder':
        arch = str(rng.choice(['cont', 'cls'])); g = int(rng.integers(1, 3000)); e = int(rng.integers(1, 70)); n = int(rng.integers(1, 700))
        dtype = str(rng.choice(['f32', 'f16x2', 'f16']))
        m = synth.synth_csr(n, g, density=float(rng.uniform(0.01, 0.6)), seed=int(rng.integers(1 << 30)))
        ws = eo.glorot_weights(arch, e, g, seed=int(rng.integers(1 << 30)), randomize_bn=True, bias_scale=0.2)
        enc = cc.HipEncoder(e, g, arch, batch_rows=int(rng.integers(1, 400)), dtype=dtype)
        enc.set_weights(ws)
        out = enc.predict(m)
        truth = eo.forward_f16(arch, ws, m.to_dense()) if dtype == 'f16' else eo.forward(arch, ws, m.to_dense())
        rel = float(np.max(np.abs(out - truth) / np.abs(truth)))
        ok = rel <= 1e-4 and np.array_equal(enc.predict(m), out)
        desc = (what, arch, g, e, n, dtype, rel)
    elif what == 'sparse':
        # sparse first layer: densities from 0.2 % to 60 % (windows of > 128 entries per 16 rows x 64 columns), ragged
        # shapes, arbitrary contiguous row ranges, non-integer values
        arch = str(rng.choice(['cont', 'cls'])); g = int(rng.integers(1, 4000)); e = int(rng.integers(1, 70)); n = int(rng.integers(1, 1500))
        dens = float(10.0 ** rng.uniform(-2.7, -0.2))
        m = synth.synth_csr(n, g, density=dens, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.3:
            m = cc.CellMatrix(m.indptr, m.indices, (m.values * rng.uniform(0.01, 3.0, len(m.values))).astype(np.float32), g)
        ws = eo.glorot_weights(arch, e, g, seed=int(rng.integers(1 << 30)), randomize_bn=True, bias_scale=0.2)
        enc = cc.HipEncoder(e, g, arch, sparse=True, batch_rows=[None, 16, 100][int(rng.integers(0, 3))])   # sparse passes of 65536 / 2048 rows
        enc.set_weights(ws)
        out = enc.predict(m)
        truth = eo.forward(arch, ws, m.to_dense())
        rel = float(np.max(np.abs(out - truth) / np.abs(truth)))
        lo = int(rng.integers(0, n)); cnt = int(rng.integers(0, n - lo + 1))
        part = torch.empty((max(cnt, 1), e), dtype=torch.float32, device='cuda')
        enc.predict_to_device(m, part.data_ptr(), lo, cnt)
        ok = rel <= 1e-4 and np.array_equal(enc.predict(m), out) and np.array_equal(part.cpu().numpy()[:cnt], out[lo:lo + cnt])
        desc = (what, arch, g, e, n, round(dens, 4), rel, lo, cnt)
    elif what == 'dense_host':
        g = int(rng.integers(1, 1200)); e = int(rng.integers(1, 40)); n = int(rng.integers(1, 9000))
        dtype = str(rng.choice(['f32', 'f16x2', 'f16']))
        m = synth.synth_csr(n, g, density=0.05, seed=int(rng.integers(1 << 30)))
        enc = cc.HipEncoder(e, g, str(rng.choice(['cont', 'cls'])), seed=int(rng.integers(1 << 30)), dtype=dtype,
                            batch_rows=int(rng.choice([64, 1000, 2048, 8192])))
        ok = np.array_equal(enc.predict(m.to_dense()), enc.predict(m))
        desc = (what, g, e, n, dtype)
    else:
        g = int(rng.integers(50, 2500)); e = int(rng.integers(2, 40)); n = int(rng.integers(50, 3000))
        m = synth.synth_csr(n, g, density=0.05, seed=int(rng.integers(1 << 30)))
        enc = cc.HipEncoder(e, g, 'cont', seed=int(rng.integers(1 << 30)), batch_rows=int(rng.integers(16, 1024)))
        full = enc.predict(m)
        lo = int(rng.integers(0, n - 1)); hi = int(rng.integers(lo + 1, n + 1))
        ok = np.array_equal(enc.predict(m.take(np.arange(lo, hi))), full[lo:hi])
        desc = (what, g, e, n, lo, hi)
    cases += 1
    if not ok:
        print('MISMATCH', desc, flush=True)
        sys.exit(1)
    if cases % 10 == 0:
        print(f'{cases} cases ok, last {desc}', flush=True)
print('soak done:', cases, 'cases ok', flush=True)
