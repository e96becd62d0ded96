"""Worker of tests/test_dist_gloo.py: one rank of the cell-sharded hot path on CPU.

The compute backend here is the ORACLE (tests only): what is under test is the
sharding / all-gather / global-index logic of cellcomm_amd.pipeline, which is
the same code the GPU ranks run with HipBackend + RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cellcomm_amd import pipeline, synth  # noqa: E402
from oracle import encoder_oracle as eo   # noqa: E402
from oracle import knn_oracle as ko       # noqa: E402


class OracleEncoder:
    def __init__(self, arch, e, g, seed):
        self.arch, self.encoding_size = arch, e
        self.weights = eo.glorot_weights(arch, e, g, seed, randomize_bn=True, bias_scale=0.1)


class OracleBackend:
    device = 'cpu'
    torch = torch

    def alloc(self, rows, width):
        return torch.empty((rows, width), dtype=torch.float32)

    def encode_into(self, encoder, matrix, out):
        n = matrix.shape[0]
        if n:
            out[:n] = torch.from_numpy(eo.predict_like_keras(encoder.arch, encoder.weights, matrix.to_dense()))

    def knn(self, enc_all, n_total, q_begin, n_q, k, include_self):
        return ko.knn(enc_all.numpy(), k, include_self, q_begin, n_q)


def main():
    out_dir, n, g, e, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    full = synth.synth_csr(n, g, density=0.1, seed=99)
    lo, hi = pipeline.shard_bounds(n, world, rank)
    local = full.take(np.arange(lo, hi))
    res = pipeline.encode_and_knn(OracleEncoder('cont', e, g, 5), local, k, backend=OracleBackend())
    # with the total known up front (no block-size exchange) the outcome is the same; `timings` collects this rank's
    # host-clock stages (what bench.py turns into stages_by_rank / allgather_ms)
    timings = {}
    res2 = pipeline.encode_and_knn(OracleEncoder('cont', e, g, 5), local, k, backend=OracleBackend(), n_total=n, timings=timings)
    assert np.array_equal(res2['idx'], res['idx']) and np.array_equal(res2['dist'], res['dist'])
    assert res2['q_begin'] == res['q_begin'] and res2['n_total'] == n
    assert set(timings) == {'encode_ms', 'allgather_ms', 'knn_ms'} and all(v >= 0 for v in timings.values())
    pipeline.encode_and_knn(OracleEncoder('cont', e, g, 5), local, k, backend=OracleBackend(), n_total=n, timings=timings)
    assert all(v > 0 for v in timings.values())           # accumulated over calls
    try:
        pipeline.encode_and_knn(OracleEncoder('cont', e, g, 5), local, k, backend=OracleBackend(), n_total=n + world)
        raise SystemExit('a wrong n_total went unnoticed')
    except ValueError:
        pass
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), enc=res['encodings'].numpy(), idx=res['idx'],
             dist=res['dist'], q_begin=res['q_begin'], n_total=res['n_total'], lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
