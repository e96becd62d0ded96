"""BASELINE configs[3] and configs[4] at their FULL size as single calls on one GPU.

The two 8-GPU configs fit one MI355X (12 GB / 6.6 GB of CSR), so the whole job can run as ONE encode call and ONE
search with n_q = N — the sizes where 32-bit seams would live: n_q x col_cap = 5e8-entry collect buffers, 125k-workgroup
grids, 1.5e9 non-zeros, four 262 144-row chunks of the fp32 encode.  The reference cannot run either
(/root/reference/src/cell_type_training.py:12-14 pivots to a dense float64 frame: 240 GB at 1M x 30k).

Truth at this size = sampled rows / sampled queries against the oracle + size-independent properties over ALL rows
(finite, (0,1), sortedness, self-exclusion, zero-distance ties by index, idempotence across the chunk seams).
"""
import json
import os
import time

import numpy as np
import pytest

import cellcomm_amd as cc
from cellcomm_amd import synth
from cellcomm_amd.knn import knn_graph_device, last_collect_norm_form, last_fallback_count, last_full_scan_count
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def max_rel(got, truth):
    return float(np.max(np.abs(got.astype(np.float64) - truth) / np.abs(truth)))


def dense_rows(keep, rows, g):
    """The sampled rows of a device-resident CSR as a dense host block (the CSR itself never leaves the device)."""
    import torch
    indptr, indices, values = keep
    ip = indptr[torch.as_tensor(np.concatenate([rows, rows + 1]), device=indptr.device)].cpu().numpy()
    out = np.zeros((len(rows), g), dtype=np.float32)
    for o in range(len(rows)):
        lo, hi = int(ip[o]), int(ip[len(rows) + o])
        out[o, indices[lo:hi].cpu().numpy()] = values[lo:hi].cpu().numpy()
    return out


class FreeMemoryWatch:
    """Lowest free device memory seen while the block runs (sampled every few ms from a second thread: the library
    calls release the GIL) — the high-water mark of a call includes scratch that is gone again when it returns."""

    def __init__(self):
        import threading
        import torch
        self.torch, self.low = torch, torch.cuda.mem_get_info()[0]
        self._stop, self._thread = threading.Event(), threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.wait(0.004):
            self.low = min(self.low, self.torch.cuda.mem_get_info()[0])

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join()
        self.low = min(self.low, self.torch.cuda.mem_get_info()[0])


def full_config(n, g, e, k, dtype, seed, seams, report_name):
    import torch
    torch.cuda.empty_cache()
    free_start, total_mem = torch.cuda.mem_get_info()   # absolute figures below: whatever earlier tests of the session left cached is returned first
    t0 = time.perf_counter()
    m, keep = synth.synth_csr_device(n, g, seed=seed)
    t_synth = time.perf_counter() - t0
    torch.cuda.empty_cache()                                                # the generator's temporaries
    nnz = int(keep[0][-1].item())
    watch = FreeMemoryWatch()
    watch.__enter__()
    enc = cc.HipEncoder(e, g, 'cont', seed=4321, dtype=dtype)
    out_dev = torch.empty((n, e), dtype=torch.float32, device='cuda')
    enc.predict_to_device(m, out_dev.data_ptr(), 0, n)                      # warm (packing, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc.predict_to_device(m, out_dev.data_ptr(), 0, n)                      # ONE call, all n rows
    t_enc = time.perf_counter() - t0
    assert enc.fp32_fallback_passes() == 0
    out = out_dev.cpu().numpy()
    assert out.shape == (n, e) and np.all(np.isfinite(out)) and np.all(out > 0) and np.all(out < 1)

    # sampled rows against the float64 oracle (of this dtype's arithmetic)
    rows = np.sort(np.concatenate([np.random.default_rng(seed).choice(n, 28, replace=False), [0, seams[0][0] + 1, n - 2, n - 1]]))
    dense = dense_rows(keep, rows, g)
    forward = eo.forward_f16 if dtype == 'f16' else eo.forward
    rel = max_rel(out[rows], forward('cont', enc.get_weights(), dense))
    assert rel <= 1e-4, rel
    # ranges across the chunk / pass seams of the long call, encoded alone: the same bits
    piece = torch.empty((max(hi - lo for lo, hi in seams), e), dtype=torch.float32, device='cuda')
    for lo, hi in seams:
        enc.predict_to_device(m, piece.data_ptr(), lo, hi - lo)
        assert np.array_equal(piece[:hi - lo].cpu().numpy(), out[lo:hi]), (lo, hi)
    free_after_encode = torch.cuda.mem_get_info()[0]

    # ONE search with n_q = N on the real encodings
    idx, dist = knn_graph_device(out_dev.data_ptr(), n, e, k)               # warm (scratch allocation)
    idx, dist = idx.copy(), dist.copy()
    t0 = time.perf_counter()
    again, again_d = knn_graph_device(out_dev.data_ptr(), n, e, k)
    t_knn = time.perf_counter() - t0
    fallback, full_scan = last_fallback_count(), last_full_scan_count()
    free_after_knn = torch.cuda.mem_get_info()[0]
    watch.__exit__()
    assert np.array_equal(again, idx) and np.array_equal(again_d, dist)     # idempotent, bit for bit
    picks = np.concatenate([np.random.default_rng(seed + 1).choice(n, 28, replace=False), [0, 31, n // 2, n - 1]])
    for q in picks:
        wi, wd = ko.knn(out, k, q_begin=int(q), n_q=1)
        assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0]), int(q)
    assert np.all(idx >= 0) and np.all(idx < n) and np.all(idx != np.arange(n, dtype=np.int64)[:, None])
    assert np.all(np.diff(dist, axis=1) >= 0)
    zero_ties = (dist[:, 1:] == 0) & (dist[:, :-1] == 0)
    assert np.all(np.diff(idx, axis=1)[zero_ties] > 0)
    for row in idx[picks]:                                                   # no candidate listed twice
        assert len(set(row.tolist())) == k

    report = {'config': report_name, 'cells': n, 'genes': g, 'latent': e, 'k': k, 'dtype': dtype, 'nnz': nnz,
              'synth_s': round(t_synth, 2), 'encode_ms_one_call': round(t_enc * 1e3, 1),
              'encode_cells_per_s': round(n / t_enc), 'knn_ms_one_call_incl_copy_out': round(t_knn * 1e3, 1),
              'knn_fallback_queries': fallback, 'knn_full_scan_queries': full_scan,
              'knn_collect_norm_form': last_collect_norm_form(),    # 2 = the accumulator form (the library's choice when hits are rare)
              'sampled_rows_max_rel_vs_oracle': rel, 'sampled_queries_bit_exact': len(picks),
              'device_GiB_used_after_encode': round((total_mem - free_after_encode) / 2**30, 2),
              'device_GiB_used_after_knn': round((total_mem - free_after_knn) / 2**30, 2),
              'device_GiB_used_high_water_mark': round((total_mem - watch.low) / 2**30, 2),
              'device_GiB_used_before': round((total_mem - free_start) / 2**30, 2),
              'csr_GiB': round((nnz * 8 + (n + 1) * 8) / 2**30, 2)}
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', f'full_{report_name}.json'), 'w') as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report), flush=True)
    del enc, m, keep, out_dev
    torch.cuda.empty_cache()
    return report


def test_config4_full_size_1m_cells_one_encode_call_one_search(hip, oracle_built):
    # configs[3]: 1M cells x 30k genes, latent 64, k = 15 — fp32, 123 passes of 8192 rows in 4 chunks of 32 passes
    # (api.hip, encode_csr_dev), then one search of 1M queries against 1M candidates
    rep = full_config(1000000, 30000, 64, 15, 'f32', seed=404,
                      seams=[(262144 - 300, 262144 + 300), (3 * 262144 - 17, 3 * 262144 + 8192 + 5), (999000, 1000000)],
                      report_name='configs3_1m')
    assert rep['nnz'] < 2**31


def test_config5_full_size_500k_cells_fp16_k50(hip, oracle_built):
    # configs[4]: 500k cells x 33k genes, fp16 operands, k = 50 — 8 passes of 65536 rows, one search of 500k x 500k
    full_config(500000, 33000, 64, 50, 'f16', seed=505,
                seams=[(65536 - 300, 65536 + 300), (7 * 65536 - 9, 7 * 65536 + 500), (499000, 500000)],
                report_name='configs4_500k')
