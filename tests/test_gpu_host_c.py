"""The C ABI from a plain C host (examples/host_c.c, built with gcc): no Python or PyTorch in that process.
Its seeded inputs are rebuilt here, pushed through the Python surface and the oracle, and the checksums
it prints must match."""
import os
import subprocess

import numpy as np
import pytest

import cellcomm_amd as cc
from conftest import ROOT
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu

MASK = (1 << 64) - 1


class Lcg:
    def __init__(self, seed):
        self.s = seed

    def next(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & MASK
        return self.s >> 32

    def unit(self):
        return np.float32(self.next() >> 8) * np.float32(1.0 / 16777216.0)


def rebuild_inputs(n, g):
    r = Lcg(12345)
    indptr, indices, vals = [0], [], []
    for _ in range(n):
        c = r.next() % 16
        while c < g:
            indices.append(c)
            vals.append(float(1 + r.next() % 4))
            c += 1 + r.next() % 31
        indptr.append(len(indices))
    return cc.CellMatrix(np.array(indptr, np.int64), np.array(indices, np.int32), np.array(vals, np.float32), g)


def rebuild_weights(enc):
    r = Lcg(777)
    ws = []
    for w in enc.get_weights():
        if w.ndim == 1:
            ws.append(np.array([np.float32(0.5) + r.unit() for _ in range(w.shape[0])], np.float32))
        else:
            lim = np.sqrt(np.float32(6.0) / np.float32(w.shape[0] + w.shape[1]), dtype=np.float32)
            flat = [(np.float32(2.0) * r.unit() - np.float32(1.0)) * lim for _ in range(w.size)]
            ws.append(np.array(flat, np.float32).reshape(w.shape))
    return ws


def test_plain_c_host_matches_python_surface_and_oracle(hip, oracle_built, tmp_path):
    n, g, e, k = 500, 1500, 16, 15
    exe = tmp_path / 'host_c'
    libdir = os.path.join(ROOT, 'cellcomm_amd')
    subprocess.run(['gcc', '-O2', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'host_c.c'), '-o',
                    str(exe), '-L' + libdir, '-lcellcomm_hip', '-Wl,-rpath,' + libdir, '-lm'], check=True)
    proc = subprocess.run([str(exe), str(n), str(g), str(e), str(k)], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    got = dict(line.split(None, 1) for line in proc.stdout.strip().splitlines()[1:])

    m = rebuild_inputs(n, g)
    enc = cc.HipEncoder(e, g, 'cont', seed=0)
    ws = rebuild_weights(enc)
    enc.set_weights(ws)
    z = enc.predict(m)
    assert abs(float(np.sum(z.astype(np.float64))) - float(got['encodings_sum'])) <= 1e-7 * abs(float(got['encodings_sum']))
    truth = eo.forward('cont', ws, m.to_dense())
    assert float(np.max(np.abs(z - truth) / np.abs(truth))) <= 1e-4
    idx, dist = ko.knn(z, k)
    h = 1469598103934665603
    for v in idx.ravel():
        h = ((h ^ (int(v) & 0xffffffff)) * 1099511628211) & MASK
    assert f'{h:016x}' == got['knn_idx_fnv'].strip()
    assert abs(float(np.sum(dist.astype(np.float64))) - float(got['knn_dist_sum'])) <= 1e-7 * float(got['knn_dist_sum'])
