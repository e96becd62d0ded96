"""N > 1 with the real HIP backend: ranks share the one GPU of the test box, gloo carries the
all-gather.  Checks shard encodings == single-process encodings (bit-exact: a row's encoding does
not depend on which rank or batch it sits in) and global neighbour ids == oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cellcomm_amd as cc
from conftest import ROOT
from cellcomm_amd import synth
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world,n,mode', [(2, 1501, 'f32'), (3, 700, 'f32'), (2, 1501, 'sparse'), (2, 1501, 'f16x2'),
                                          (2, 900, 'f16')])
def test_sharded_hip_path(hip, oracle_built, tmp_path, world, n, mode):
    """Every encoder mode under the sharded pipeline: a shard's encodings are the single-process encodings of the same
    mode, bit for bit (a row's result depends neither on the rank nor on the pass it sits in)."""
    g, e, k = 2000, 32, 15
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', '_gloo_gpu_worker.py'), str(tmp_path), str(n), str(g), str(e), str(k), mode]
    proc = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'), capture_output=True,
                          text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    full = synth.synth_csr(n, g, density=0.05, seed=99)
    want_enc = cc.HipEncoder(e, g, 'cont', seed=5, sparse=mode == 'sparse',
                             dtype=mode if mode.startswith('f16') else 'f32').predict(full)
    want_i, want_d = ko.knn(want_enc, k)
    covered = 0
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        lo, hi = int(z['lo']), int(z['hi'])
        assert int(z['n_total']) == n and int(z['q_begin']) == lo
        assert np.array_equal(z['enc'], want_enc)
        assert np.array_equal(z['idx'], want_i[lo:hi]) and np.array_equal(z['dist'], want_d[lo:hi])
        covered += hi - lo
    assert covered == n
