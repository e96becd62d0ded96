"""The C-ABI exchange (cc_comm_*, cc_allgather_f32, cc_knn_l2_sharded): RCCL bound at run time.
A one-GPU box can only host a one-rank communicator (RCCL refuses two ranks on one device), so this
pins the plumbing — unique id, init, the collective on the library stream, shard bookkeeping, errors —
and the N > 1 arithmetic of the same code path is covered over gloo (test_gpu_dist.py, test_dist_gloo.py)."""
import numpy as np
import pytest

import cellcomm_amd as cc
from cellcomm_amd import comm as cccomm, pipeline, synth
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm1(hip):
    c = cccomm.NativeComm(1, 0, cccomm.make_unique_id())
    yield c
    c.close()


def test_unique_id_and_bad_arguments(hip):
    a, b = cccomm.make_unique_id(), cccomm.make_unique_id()
    assert len(a) == 128 and a != b
    with pytest.raises(ValueError):
        cccomm.NativeComm(1, 0, b'short')
    with pytest.raises(ValueError):
        cccomm.NativeComm(2, 2, a)            # rank outside the communicator


def test_allgather_and_sharded_knn_single_rank(comm1, oracle_built):
    import torch
    x = np.random.default_rng(1).random((3000, 32), dtype=np.float32)
    t = torch.from_numpy(x).cuda()
    out = torch.zeros_like(t)
    torch.cuda.synchronize()
    comm1.all_gather(t.data_ptr(), t.numel(), out.data_ptr())
    assert torch.equal(out, t)
    idx, dist = comm1.knn_sharded(t.data_ptr(), 3000, 3000, 32, 15)
    want_i, want_d = ko.knn(x, 15)
    assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d)
    with pytest.raises(ValueError, match='must hold 3000 of 3000 rows'):
        comm1.knn_sharded(t.data_ptr(), 2999, 3000, 32, 15)
    with pytest.raises(ValueError, match='Expected n_neighbors <= n_samples_fit'):
        comm1.knn_sharded(t.data_ptr(), 10, 10, 32, 10)


def test_pipeline_with_native_comm(comm1, oracle_built):
    n, g, e, k = 900, 2000, 32, 15
    m = synth.synth_csr(n, g, density=0.05, seed=7)
    enc = cc.HipEncoder(e, g, 'cont', seed=3)
    res = pipeline.encode_and_knn_native(enc, m, k, comm1, n)
    ref = pipeline.encode_and_knn(enc, m, k)
    assert np.array_equal(res['encodings'].cpu().numpy(), ref['encodings'].cpu().numpy())
    want_i, want_d = ko.knn(ref['encodings'].cpu().numpy(), k)
    assert np.array_equal(res['idx'], want_i) and np.array_equal(res['dist'], want_d)
    with pytest.raises(ValueError, match='must hold rows'):
        pipeline.encode_and_knn_native(enc, m, k, comm1, n + 1)
