"""Performance floors for the hot kernels, at 0.92 x the rates of the newest TRACKED bench line
(`profiles/rNN_bench_line_under_rocprof.json`; boxes differ by ~3 %), so that a 10 % regression fails and a future
improvement raises the floors by committing its profile, without an edit here (the old fixed floors
let a 20-25 % regression of the headline kernels pass).  They exist to catch silent regressions — a register spill or a
different register assignment cost the layer-1 GEMM 11 % without changing a single result — not to certify the roofline
numbers (bench.py and profiles/ do that).  Each figure is the best of up to three timings: the floor is about the code,
not about a noisy moment of the box."""
import glob
import json
import os
import re

import pytest

import cellcomm_amd as cc
from cellcomm_amd import _lib, synth
from cellcomm_amd.knn import knn_graph_device
from conftest import ROOT

pytestmark = pytest.mark.gpu

FRACTION = 0.92
KNN_SLACK = 1.25          # the test searches uniform random points, the tracked line real encodings (3.4 -> 4.2 ms)


def tracked_line():
    files = glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_line_under_rocprof.json'))
    newest = max(files, key=lambda f: int(re.search(r'r(\d+)_', os.path.basename(f)).group(1)))
    return os.path.basename(newest), json.load(open(newest))


def floors():
    name, line = tracked_line()
    st = line['stages_rank0']
    knn_ms = sum(st[k]['ms_per_step'] for k in ('knn_pack', 'knn_select', 'knn_rerank'))
    return name, {
        'gemm_l1': FRACTION * st['gemm_l1']['rate'], 'gemm_l2': FRACTION * st['gemm_l2']['rate'],
        'gather': FRACTION * st['gather']['rate'], 'knn_ms': KNN_SLACK * knn_ms,
        'f16': FRACTION * line['alt_f16']['layer1_algorithmic_tflops'] * 1e12,
        'f16x2': FRACTION * line['alt']['layer1_algorithmic_tflops'] * 1e12,
        'sparse': FRACTION * line['alt_sparse']['roofline']['fp32_fma_tflops'] * 1e12,
    }


def test_floors_follow_the_tracked_profile():
    name, f = floors()
    assert re.match(r'r\d+_bench_line_under_rocprof\.json', name)
    # sanity of the tracked file itself (a truncated or mis-keyed line must not turn the floors off); the gather of the
    # tracked runs has been 5.98-6.20 TB/s under rocprofv3, so its bound sits below 0.92 x the lowest of them
    assert f['gemm_l1'] >= 125e12 and f['gemm_l2'] >= 125e12 and f['gather'] >= 5.3e12 and f['knn_ms'] <= 4.4
    assert f['f16'] >= 1250e12 and f['f16x2'] >= 720e12 and f['sparse'] >= 31e12


def best_of(measure, good_enough, tries=3):
    best = None
    for _ in range(tries):
        got = measure()
        best = got if best is None else {k: max(best[k], got[k]) for k in got}
        if good_enough(best):
            break
    return best


def test_kernel_rates_stay_above_their_floors(hip):
    import torch
    name, f = floors()
    n, g, e = 8192, 30000, 64
    m, _ = synth.synth_csr_device(n, g, seed=5)
    enc = cc.HipEncoder(e, g, 'cont', seed=6)
    out = torch.empty((n, e), device='cuda')
    enc.predict_to_device(m, out.data_ptr(), 0, 8192)                     # warm-up: packs weights, allocates workspaces
    x = torch.rand((100000, 64), device='cuda')
    torch.cuda.synchronize()
    knn_graph_device(x.data_ptr(), 100000, 64, 15)                        # ... and the search scratch (cc_shutdown in an earlier test released it)

    def measure():
        _lib.profile_enable(True)
        _lib.profile_reset()
        try:
            for _ in range(3):
                enc.predict_to_device(m, out.data_ptr(), 0, 8192)         # one 8192-row pass: the launch shape of the bench
            knn_graph_device(x.data_ptr(), 100000, 64, 15)
            prof = _lib.profile_snapshot()
        finally:
            _lib.profile_enable(False)
        rate = {k: v['work'] / (v['ms'] * 1e-3) for k, v in prof.items() if v['ms'] > 0}
        rate['knn_inv_ms'] = 1.0 / sum(prof[k]['ms'] for k in ('knn_pack', 'knn_select', 'knn_rerank'))
        return {k: rate[k] for k in ('gemm_l1', 'gemm_l2', 'gather', 'knn_inv_ms')}

    def ok(r):
        return all(r[k] >= f[k] for k in ('gemm_l1', 'gemm_l2', 'gather')) and 1.0 / r['knn_inv_ms'] <= f['knn_ms']

    r = best_of(measure, ok)
    for k in ('gemm_l1', 'gemm_l2'):
        assert r[k] >= f[k], f"{k} at {r[k] / 1e12:.1f} TFLOP/s, floor {f[k] / 1e12:.1f} = {FRACTION} x {name}"
    assert r['gather'] >= f['gather'], f"gather at {r['gather'] / 1e12:.2f} TB/s, floor {f['gather'] / 1e12:.2f} ({name})"
    knn_ms = 1.0 / r['knn_inv_ms']
    assert knn_ms <= f['knn_ms'], f"kNN kernels of a 100k x 100k call took {knn_ms:.2f} ms, ceiling {f['knn_ms']:.2f} ({name})"


def test_opt_in_paths_stay_above_their_floors(hip):
    """The sparse first layer and the fp16 matrix-core kernels on one 32768-row call at C3's width."""
    import torch
    name, f = floors()
    n, g, e = 32768, 30000, 64
    m, _ = synth.synth_csr_device(n, g, seed=5)
    out = torch.empty((n, e), device='cuda')
    for mode, stage in (('sparse', 'spmm_l1'), ('f16', 'gemm_l1'), ('f16x2', 'gemm_l1')):
        enc = cc.HipEncoder(e, g, 'cont', seed=6, sparse=(mode == 'sparse'), dtype=mode if mode.startswith('f16') else 'f32')
        enc.predict_to_device(m, out.data_ptr(), 0, n)                    # warm-up: packs weights, builds the entry stream

        def measure():
            _lib.profile_enable(True)
            _lib.profile_reset()
            try:
                for _ in range(2):
                    enc.predict_to_device(m, out.data_ptr(), 0, n)
                prof = _lib.profile_snapshot()
            finally:
                _lib.profile_enable(False)
            return {'rate': prof[stage]['work'] / (prof[stage]['ms'] * 1e-3)}

        r = best_of(measure, lambda b: b['rate'] >= f[mode])
        assert r['rate'] >= f[mode], f"{mode}: {stage} at {r['rate'] / 1e12:.1f} TFLOP/s, floor {f[mode] / 1e12:.1f} = {FRACTION} x {name}"
        del enc
