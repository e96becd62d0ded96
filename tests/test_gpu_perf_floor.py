"""Performance floors for the hot kernels (generous: ~75 % of what round 1 measured, so that a throttled box does not trip them).  They exist to catch silent
regressions — a register spill or a different register assignment cost the layer-1 GEMM 11 % without changing a
single result — not to certify the roofline numbers (bench.py and profiles/ do that)."""
import numpy as np
import pytest

import cellcomm_amd as cc
from cellcomm_amd import _lib, synth
from cellcomm_amd.knn import knn_graph_device

pytestmark = pytest.mark.gpu


def test_kernel_rates_stay_above_their_floors(hip):
    import torch
    n, g, e = 8192, 30000, 64
    m, _ = synth.synth_csr_device(n, g, seed=5)
    enc = cc.HipEncoder(e, g, 'cont', seed=6)
    out = torch.empty((n, e), device='cuda')
    enc.predict_to_device(m, out.data_ptr(), 0, 8192)                     # warm-up: packs weights, allocates workspaces
    x = torch.rand((100000, 64), device='cuda')
    torch.cuda.synchronize()
    knn_graph_device(x.data_ptr(), 100000, 64, 15)                        # ... and the search scratch (cc_shutdown in an earlier test released it)
    _lib.profile_enable(True)
    _lib.profile_reset()
    try:
        for _ in range(3):
            enc.predict_to_device(m, out.data_ptr(), 0, 8192)             # one 8192-row pass: the launch shape of the bench
        knn_graph_device(x.data_ptr(), 100000, 64, 15)
        prof = _lib.profile_snapshot()
    finally:
        _lib.profile_enable(False)
    rate = {k: v['work'] / (v['ms'] * 1e-3) for k, v in prof.items() if v['ms'] > 0}
    assert rate['gemm_l1'] >= 105e12, f"layer-1 GEMM at {rate['gemm_l1'] / 1e12:.1f} TFLOP/s (round 1: 139)"
    assert rate['gemm_l2'] >= 105e12, f"layer-2 GEMM at {rate['gemm_l2'] / 1e12:.1f} TFLOP/s (round 1: 139)"
    assert rate['gather'] >= 4.0e12, f"gather at {rate['gather'] / 1e12:.2f} TB/s (round 1: 6.2)"
    knn_ms = sum(prof[k]['ms'] for k in ('knn_pack', 'knn_select', 'knn_rerank'))
    assert knn_ms <= 7.0, f'kNN kernels of a 100k x 100k call took {knn_ms:.1f} ms (round 1: 8.2, round 2: 3.6)'


def test_opt_in_paths_stay_above_their_floors(hip):
    """The sparse first layer and the fp16 matrix-core kernels (35 TFLOP/s on the non-zero products; one fp16 plane
    1.41-1.44 PFLOP/s at 32768 rows since the ping-pong K loop of round 3, two planes 0.80 algorithmic)."""
    import torch
    n, g, e = 32768, 30000, 64
    m, _ = synth.synth_csr_device(n, g, seed=5)
    out = torch.empty((n, e), device='cuda')
    # round 3: f16 1408-1437 TFLOP/s, f16x2 801 algorithmic (= 1603 executed) at this shape (1281-1346 / 715-762 before
    # the ping-pong K loop)
    floors = {'sparse': ('spmm_l1', 20e12), 'f16': ('gemm_l1', 1100e12), 'f16x2': ('gemm_l1', 620e12)}
    for mode, (stage, floor) in floors.items():
        enc = cc.HipEncoder(e, g, 'cont', seed=6, sparse=(mode == 'sparse'), dtype=mode if mode.startswith('f16') else 'f32')
        enc.predict_to_device(m, out.data_ptr(), 0, n)                    # warm-up: packs weights, builds the entry stream
        _lib.profile_enable(True)
        _lib.profile_reset()
        try:
            for _ in range(2):
                enc.predict_to_device(m, out.data_ptr(), 0, n)
            prof = _lib.profile_snapshot()
        finally:
            _lib.profile_enable(False)
        rate = prof[stage]['work'] / (prof[stage]['ms'] * 1e-3)
        assert rate >= floor, f'{mode}: {stage} at {rate / 1e12:.1f} TFLOP/s, floor {floor / 1e12:.0f}'
        del enc
