"""Robustness of the C ABI: state lives in handles, error paths do not leak
or corrupt later calls, the CSR canonical form is checked on the device, out-of-range fp16 inputs fall back
to fp32, the dense-host pipeline equals the CSR path, shutdown / re-init works."""
import ctypes

import numpy as np
import pytest
import torch

import cellcomm_amd as cc
from cellcomm_amd import _lib, synth
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def max_rel(got, truth):
    return float(np.max(np.abs(got.astype(np.float64) - truth) / np.abs(truth)))


def test_two_encoders_of_different_width_interleaved(hip):
    """Workspaces, packed weights and pass buffers belong to each encoder handle: alternating calls between
    encoders of different gene counts, architectures and dtypes must not disturb each other."""
    specs = [('cont', 1500, 16, 'f32'), ('cls', 2300, 8, 'f32'), ('cont', 2300, 12, 'f16x2'), ('cont', 640, 5, 'f16')]
    encs, mats, truth = [], [], []
    for i, (arch, g, e, dtype) in enumerate(specs):
        ws = eo.glorot_weights(arch, e, g, seed=20 + i, randomize_bn=True, bias_scale=0.05)
        enc = cc.HipEncoder(e, g, arch, dtype=dtype, batch_rows=256)
        enc.set_weights(ws)
        m = synth.synth_csr(700 + 13 * i, g, seed=40 + i)
        encs.append(enc)
        mats.append(m)
        dense = m.to_dense()
        truth.append(eo.forward_f16(arch, ws, dense) if dtype == 'f16' else eo.forward(arch, ws, dense))
    first = [None] * len(specs)
    for rnd in range(3):
        for i in (0, 1, 2, 3, 2, 0, 3, 1):
            out = encs[i].predict(mats[i]) if (rnd + i) % 2 == 0 else encs[i].predict(mats[i].to_dense())
            assert max_rel(out, truth[i]) <= RTOL
            if first[i] is None:
                first[i] = out
            assert np.array_equal(out, first[i])        # same bits on every call, sparse or dense input


def test_dense_host_pipeline_equals_csr_path(hip):
    """cc_encode_dense runs passes of <= 4096 rows through two slots on two streams (copy of pass i+1 under the
    kernels of pass i): 5 passes incl. a ragged last one, all three dtypes."""
    g, e, n = 1024, 16, 4096 * 4 + 777
    m = synth.synth_csr(n, g, seed=3)
    dense = m.to_dense()
    for dtype in ('f32', 'f16x2', 'f16'):
        enc = cc.HipEncoder(e, g, 'cont', seed=8, dtype=dtype)
        a = enc.predict(m)
        b = enc.predict(dense)
        assert np.array_equal(a, b), dtype
        pinned = torch.from_numpy(dense).pin_memory()            # fully asynchronous copies
        assert np.array_equal(enc.predict(pinned.numpy()), a), dtype
    enc = cc.HipEncoder(e, g, 'cont', seed=8)
    assert max_rel(enc.predict(dense), eo.forward('cont', enc.get_weights(), dense)) <= RTOL
    assert enc.predict(dense[:0]).shape == (0, e)
    assert np.array_equal(enc.predict(dense[:1]), enc.predict(m)[:1])


def test_csr_canonical_form_is_checked_on_the_device(hip):
    lib = _lib.lib()

    def upload(indptr, idx, g):
        indptr = np.asarray(indptr, dtype=np.int64)
        idx = np.asarray(idx, dtype=np.int32)
        vals = np.ones(len(idx), dtype=np.float32)
        h = ctypes.c_void_p()
        rc = lib.cc_csr_upload(indptr.ctypes.data, idx.ctypes.data, vals.ctypes.data, len(indptr) - 1, len(idx), g,
                               ctypes.byref(h))
        msg = lib.cc_last_error().decode()
        if rc == 0:
            lib.cc_csr_free(h)
        return rc, msg

    assert upload([0, 2, 4], [0, 3, 1, 2], 8)[0] == 0
    rc, msg = upload([0, 2, 5], [0, 3, 2, 2, 4], 8)
    assert rc == 1 and 'columns of row 1 not strictly increasing' in msg
    rc, msg = upload([0, 2, 4, 5], [0, 3, 1, 9, 2], 8)
    assert rc == 1 and 'row 1' in msg and 'outside [0,8)' in msg
    rc, msg = upload([0, 3, 2, 5], [0, 1, 2, 3, 4], 8)
    assert rc == 1 and 'indptr not monotone' in msg and 'row 1' in msg
    rc, msg = upload([0, 2, 4], [0, 3, 1, 2, 7], 8)               # indptr[-1] != nnz
    assert rc == 1 and 'indptr must start at 0 and end at nnz' in msg
    # the first offending row is reported, also far into a large matrix
    m = synth.synth_csr(50000, 300, seed=1)
    idx = m.indices.copy()
    bad_row = 31337
    lo, hi = int(m.indptr[bad_row]), int(m.indptr[bad_row + 1])
    assert hi - lo >= 2
    idx[lo], idx[lo + 1] = idx[lo + 1], idx[lo]
    vals = np.ones(len(idx), dtype=np.float32)
    h = ctypes.c_void_p()
    rc = lib.cc_csr_upload(m.indptr.ctypes.data, idx.ctypes.data, vals.ctypes.data, 50000, len(idx), 300, ctypes.byref(h))
    assert rc == 1 and f'columns of row {bad_row} not strictly increasing' in lib.cc_last_error().decode()
    # wrapping device arrays runs the same check
    d_ptr = torch.from_numpy(m.indptr).cuda()
    d_idx = torch.from_numpy(idx).cuda()
    d_val = torch.from_numpy(vals).cuda()
    rc = lib.cc_csr_wrap_device(d_ptr.data_ptr(), d_idx.data_ptr(), d_val.data_ptr(), 50000, len(idx), 300, ctypes.byref(h))
    assert rc == 1 and f'row {bad_row}' in lib.cc_last_error().decode()


def test_failed_calls_leave_the_library_usable(hip):
    g, e = 900, 8
    m = synth.synth_csr(400, g, seed=2)
    enc = cc.HipEncoder(e, g, 'cont', seed=1)
    want = enc.predict(m)
    lib = _lib.lib()
    out = np.empty((4, e), dtype=np.float32)
    ids = np.array([0, 1, 400, 2], dtype=np.int64)
    for _ in range(20):
        assert lib.cc_encode(enc._h, m.device_handle(), ids.ctypes.data, 0, 4, out.ctypes.data) == 1
        assert 'row id 400' in lib.cc_last_error().decode()
        dense = np.empty((4, g + 4), dtype=np.float32)
        assert lib.cc_gather_dense(m.device_handle(), ids.ctypes.data, 0, 4, g + 4, dense.ctypes.data, 0) == 1
        with pytest.raises(ValueError):
            cc.knn_graph(np.zeros((5, 3), dtype=np.float32), 9)
    assert np.array_equal(enc.predict(m), want)


def test_f16_values_outside_the_fp16_range_run_in_fp32(hip):
    """in dtype 'f16' a count above 65504 became +inf and poisoned the row (inf * 0 -> NaN)."""
    g, e, n = 800, 8, 300
    m = synth.synth_csr(n, g, seed=6)
    vals = m.values.copy()
    vals[int(m.indptr[17])] = 70000.0                        # one cell with an out-of-range count
    big = cc.CellMatrix(m.indptr, m.indices, vals, g)
    ws = eo.glorot_weights('cont', e, g, seed=4, randomize_bn=True, bias_scale=0.05)
    truth = eo.forward('cont', ws, big.to_dense())
    for dtype in ('f16', 'f16x2'):
        enc = cc.HipEncoder(e, g, 'cont', dtype=dtype, batch_rows=128)
        enc.set_weights(ws)
        out = enc.predict(big)
        assert np.isfinite(out).all()
        assert enc.fp32_fallback_passes() == 1               # only the pass that holds row 17
        assert max_rel(out[:128], truth[:128]) <= RTOL       # that pass ran exact fp32
        dense_out = enc.predict(big.to_dense())
        assert np.isfinite(dense_out).all() and enc.fp32_fallback_passes() == 1
        assert np.array_equal(dense_out[:128], out[:128])
    # in-range but inexact values (2049 is not an fp16 number): f16 rounds them as its contract says, f16x2 falls back
    vals2 = m.values.copy()
    vals2[int(m.indptr[200])] = 2049.0
    odd = cc.CellMatrix(m.indptr, m.indices, vals2, g)
    enc16 = cc.HipEncoder(e, g, 'cont', dtype='f16', batch_rows=128)
    enc16.set_weights(ws)
    out16 = enc16.predict(odd)
    assert enc16.fp32_fallback_passes() == 0
    assert max_rel(out16, eo.forward_f16('cont', ws, odd.to_dense())) <= RTOL
    encx2 = cc.HipEncoder(e, g, 'cont', dtype='f16x2', batch_rows=128)
    encx2.set_weights(ws)
    assert max_rel(encx2.predict(odd), eo.forward('cont', ws, odd.to_dense())) <= RTOL
    assert encx2.fp32_fallback_passes() == 1


def test_knn_workspace_handles_are_independent(hip, oracle_built):
    lib = _lib.lib()
    rng = np.random.default_rng(12)
    x = rng.random((3000, 24), dtype=np.float32)
    x[100:140] = x[50]                                       # exact duplicates: ties by index
    want_i, want_d = ko.knn(x, 10)
    xt = torch.from_numpy(x).cuda()
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.cc_knn_create(ctypes.byref(a)))
    _lib.check(lib.cc_knn_create(ctypes.byref(b)))
    _lib.check(lib.cc_knn_set_option(a, 0, 1))               # CC_KNN_OPT_FORCE_FALLBACK on handle a only
    _lib.check(lib.cc_knn_set_option(b, 5, 6))               # six collect splits on handle b

    def search(h, q0, nq):
        oi = np.empty((nq, 10), dtype=np.int32)
        od = np.empty((nq, 10), dtype=np.float32)
        _lib.check(lib.cc_knn_search_dev(h, xt.data_ptr(), 3000, 24, q0, nq, 10, 0, oi.ctypes.data, od.ctypes.data))
        return oi, od

    def stat(h, which):
        v = ctypes.c_int64()
        _lib.check(lib.cc_knn_stat(h, which, ctypes.byref(v)))
        return v.value

    for h in (a, b, None, a, None, b):                       # interleaved; None = the default workspace
        oi, od = search(h, 500, 2000)
        assert np.array_equal(oi, want_i[500:2500]) and np.array_equal(od, want_d[500:2500])
    assert stat(a, 0) == 2000                                # every query of `a` went through float64
    assert stat(b, 0) < 2000 and stat(None, 0) < 2000
    assert lib.cc_knn_set_option(a, 99, 1) == 1 and 'unknown kNN option' in lib.cc_last_error().decode()
    assert lib.cc_knn_set_option(b, 5, 65) == 1
    lib.cc_knn_free(a)
    lib.cc_knn_free(b)
    idx, dist = cc.knn_graph(x, 10)                          # the convenience path is untouched by the handles' options
    assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d)


def test_init_is_idempotent_and_shutdown_releases_the_context(hip):
    lib = _lib.lib()
    assert lib.cc_init(0) == 0                               # same device again: no-op
    n_dev = lib.cc_device_count()
    rc = lib.cc_init(n_dev)                                  # not a device
    assert rc == 1 and 'not available' in lib.cc_last_error().decode()
    if n_dev > 1:
        assert lib.cc_init(1) == 3                           # CC_ESTATE: bound to device 0
        assert 'cc_shutdown' in lib.cc_last_error().decode()
    g, e = 500, 4
    m = synth.synth_csr(300, g, seed=9)
    enc = cc.HipEncoder(e, g, 'cls', seed=2)
    before = enc.predict(m)
    cc.knn_graph(before, 3)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    _lib.shutdown()
    assert torch.cuda.mem_get_info()[0] >= free0             # the default kNN scratch went back
    assert lib.cc_shutdown() == 0                            # twice is fine
    out = np.empty((1, e), dtype=np.float32)
    assert lib.cc_encode(enc._h, m.device_handle(), None, 0, 1, out.ctypes.data) == 1    # "cc_init has not been called"
    _lib.ensure_init(0)
    assert np.array_equal(enc.predict(m), before)            # handles made before the shutdown keep working
    idx, _ = cc.knn_graph(before, 3)
    assert idx.shape == (300, 3)


def test_null_handles_and_buffers_are_refused(hip):
    """Every entry point that dereferences a handle or a caller buffer checks it first."""
    lib = _lib.lib()
    g, e = 400, 4
    m = synth.synth_csr(50, g, seed=1)
    enc = cc.HipEncoder(e, g, 'cont', seed=1)
    out = np.empty((4, e), dtype=np.float32)
    dense = np.zeros((4, g), dtype=np.float32)
    r, c = ctypes.c_int64(), ctypes.c_int64()
    calls = [
        (lib.cc_encode, (None, m.device_handle(), None, 0, 4, out.ctypes.data), 'encoder handle is NULL'),
        (lib.cc_encode, (enc._h, None, None, 0, 4, out.ctypes.data), 'CSR handle is NULL'),
        (lib.cc_encode, (enc._h, m.device_handle(), None, 0, 4, None), 'host_out is NULL'),
        (lib.cc_encode_dev, (enc._h, m.device_handle(), None, 0, 4, None), 'dev_out is NULL'),
        (lib.cc_encode_dense, (None, dense.ctypes.data, 4, out.ctypes.data), 'encoder handle is NULL'),
        (lib.cc_encode_dense, (enc._h, None, 4, out.ctypes.data), 'is NULL'),
        (lib.cc_gather_dense, (None, None, 0, 4, g, dense.ctypes.data, 0), 'CSR handle is NULL'),
        (lib.cc_gather_dense, (m.device_handle(), None, 0, 4, g, None, 0), 'out is NULL'),
        (lib.cc_encoder_weight_shape, (None, 0, ctypes.byref(r), ctypes.byref(c)), 'encoder handle is NULL'),
        (lib.cc_encoder_set_weights, (None, 0, dense.ctypes.data, 1, 1), 'encoder handle is NULL'),
        (lib.cc_encoder_set_low_latency, (None, 1), 'encoder handle is NULL'),
        (lib.cc_encoder_set_sparse_input, (None, 1), 'encoder handle is NULL'),
        (lib.cc_encoder_set_batch, (None, 64), 'encoder handle is NULL'),
        (lib.cc_generator_weight_shape, (None, 0, ctypes.byref(r), ctypes.byref(c)), 'generator handle is NULL'),
        (lib.cc_generate_dense, (None, dense.ctypes.data, 4, dense.ctypes.data), 'generator handle is NULL'),
        (lib.cc_mtx_shape, (None, ctypes.byref(r), ctypes.byref(c), ctypes.byref(c)), 'NULL handle'),
        (lib.cc_knn_l2, (None, 10, 8, 3, 0, out.ctypes.data, out.ctypes.data), 'host_x is NULL'),
        (lib.cc_device_name, (None, 0), 'buf is NULL'),
    ]
    for fn, args, text in calls:
        assert fn(*args) == 1, fn.__name__
        assert text in lib.cc_last_error().decode(), (fn.__name__, lib.cc_last_error().decode())
    assert lib.cc_encoder_num_weights(None) == -1 and lib.cc_generator_num_weights(None) == -1
    assert lib.cc_encoder_fp32_fallback_passes(None) == -1 and lib.cc_encoder_batch(None) == -1
    lib.cc_encoder_free(None); lib.cc_csr_free(None); lib.cc_knn_free(None); lib.cc_generator_free(None)   # no-ops
    assert enc.predict(m).shape == (50, e)


def test_wrapped_indptr_must_run_from_zero_to_nnz(hip):
    """cc_csr_wrap_device checked monotone rows but not that indptr starts at 0 and ends at nnz."""
    lib = _lib.lib()
    m = synth.synth_csr(40, 64, seed=3)
    nnz = int(m.indptr[-1])
    d_idx, d_val = torch.from_numpy(m.indices).cuda(), torch.from_numpy(m.values).cuda()
    h = ctypes.c_void_p()
    shifted = m.indptr.copy(); shifted[0] = 1
    d_ptr = torch.from_numpy(shifted).cuda()
    assert lib.cc_csr_wrap_device(d_ptr.data_ptr(), d_idx.data_ptr(), d_val.data_ptr(), 40, nnz, 64, ctypes.byref(h)) == 1
    assert 'indptr not monotone from 0 to nnz at row 0' in lib.cc_last_error().decode()
    d_ptr = torch.from_numpy(m.indptr).cuda()
    assert lib.cc_csr_wrap_device(d_ptr.data_ptr(), d_idx.data_ptr(), d_val.data_ptr(), 40, nnz + 3, 64, ctypes.byref(h)) == 1
    assert 'row 39' in lib.cc_last_error().decode()
    assert lib.cc_csr_wrap_device(d_ptr.data_ptr(), d_idx.data_ptr(), d_val.data_ptr(), 40, nnz, 64, ctypes.byref(h)) == 0
    lib.cc_csr_free(h)


def test_shutdown_then_init_reruns_the_per_context_kernel_setup(hip):
    """The 160 KiB dynamic-LDS attribute of the sparse and fp16 kernels is per library context: a
    process that has used them, shut down and initialised again runs them again; options set on the default kNN
    workspace survive the shutdown."""
    lib = _lib.lib()
    g, e = 1200, 8
    m = synth.synth_csr(600, g, seed=5)
    ws = eo.glorot_weights('cont', e, g, seed=6, randomize_bn=True, bias_scale=0.05)
    truth = eo.forward('cont', ws, m.to_dense())

    def run_all():
        outs = []
        for kw in (dict(sparse=True), dict(dtype='f16x2'), dict(dtype='f16')):
            enc = cc.HipEncoder(e, g, 'cont', **kw)
            enc.set_weights(ws)
            outs.append(enc.predict(m))
            del enc
        return outs

    first = run_all()
    assert max_rel(first[0], truth) <= RTOL and max_rel(first[1], truth) <= RTOL
    _lib.check(lib.cc_knn_set_option(None, 0, 1))            # CC_KNN_OPT_FORCE_FALLBACK on the default workspace
    m._device = None                                         # the matrix handle is freed before the shutdown
    import gc
    gc.collect()
    _lib.shutdown()
    _lib.ensure_init(0)
    again = run_all()
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    x = np.random.default_rng(2).random((500, 8), dtype=np.float32)
    try:
        cc.knn_graph(x, 3)
        assert lib.cc_knn_last_fallback_count() == 500       # the option outlived the shutdown
    finally:
        _lib.check(lib.cc_knn_set_option(None, 0, 0))


def test_rows_per_pass_follow_the_layer_widths(hip):
    """The default pass is the row count whose layer-1 / layer-2 tiles fill whole rounds of the chip (256 CUs: 768 fp32
    tiles of 128 x 128, 256 fp16 tiles of 256 x 256): C3's width keeps the constants the headline was tuned with, other
    widths move; an explicit `batch_rows` always wins.  (api.hip `default_pass_rows`.)"""
    import cellcomm_amd as cc
    assert cc.HipEncoder(64, 30000, 'cont').pass_rows == 8192            # 24 + 12 column tiles: 2 + 1 rounds
    assert cc.HipEncoder(64, 30000, 'cont', dtype='f16').pass_rows == 32768    # 12 + 6: 6 + 3 rounds
    assert cc.HipEncoder(64, 30000, 'cont', dtype='f16x2').pass_rows == 32768
    assert cc.HipEncoder(10, 30000, 'cls').pass_rows == 12288            # 8 column tiles: 96 m-tiles = one round
    assert cc.HipEncoder(32, 20000, 'cont').pass_rows == 12288           # 16 + 8: 2 + 1 rounds
    assert cc.HipEncoder(64, 33000, 'cont', dtype='f16').pass_rows == 65536    # 13 + 7: 256 m-tiles = 13 + 7 rounds
    assert cc.HipEncoder(8, 5, 'cont').pass_rows == 8192                 # degenerate widths (int(G * 0.1) == 0): the constant
    assert cc.HipEncoder(64, 33000, 'cont', dtype='f16', batch_rows=1000).pass_rows == 1000
