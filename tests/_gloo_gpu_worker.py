"""Worker of tests/test_gpu_dist.py: one rank of the cell-sharded hot path with the REAL HipBackend.
All ranks share GPU 0 (the test box has one GPU); the collective runs over gloo with host staging,
everything else — gather, encoder, kNN with global query offsets — is the product path."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cellcomm_amd as cc                         # noqa: E402
from cellcomm_amd import _lib, pipeline, synth    # noqa: E402


def main():
    out_dir, n, g, e, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    mode = sys.argv[6] if len(sys.argv) > 6 else 'f32'          # f32 | sparse | f16x2 | f16
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    _lib.ensure_init(0)
    full = synth.synth_csr(n, g, density=0.05, seed=99)
    lo, hi = pipeline.shard_bounds(n, world, rank)
    local = cc.CellMatrix(full.indptr[lo:hi + 1] - full.indptr[lo], full.indices[full.indptr[lo]:full.indptr[hi]],
                          full.values[full.indptr[lo]:full.indptr[hi]], g)
    enc = cc.HipEncoder(e, g, 'cont', seed=5, sparse=mode == 'sparse', dtype=mode if mode.startswith('f16') else 'f32')
    res = pipeline.encode_and_knn(enc, local, k)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), enc=res['encodings'].cpu().numpy(), idx=res['idx'],
             dist=res['dist'], q_begin=res['q_begin'], n_total=res['n_total'], lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
