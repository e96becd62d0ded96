#!/usr/bin/env python3
"""Device-memory leak check: create/use/free every handle type in a loop — including calls that FAIL half-way
(bad row ids, malformed CSR, wrong shapes, impossible k) — and watch hipMemGetInfo; then cc_shutdown and check
that the library context (streams, pooled events, default kNN scratch) gives its memory back too."""
import ctypes
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth, pipeline
from cellcomm_amd.generator import build_cont_generator

handle = _lib.ensure_init(0)


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def expect_failure(fn, *args):
    rc = fn(*args)
    assert rc != 0, f'{fn.__name__} should have failed'
    return rc


def failing_calls(i, m, enc):
    """Every one of these returns an error after some device work or allocation has already happened."""
    n, g = m.shape
    out = np.empty((4, enc.encoding_size), dtype=np.float32)
    bad_ids = np.array([0, 1, n + 5, 2], dtype=np.int64)
    expect_failure(handle.cc_encode, enc._h, m.device_handle(), bad_ids.ctypes.data, 0, 4, out.ctypes.data)
    dense = np.empty((4, g + 8), dtype=np.float32)
    expect_failure(handle.cc_gather_dense, m.device_handle(), bad_ids.ctypes.data, 0, 4, g + 8, dense.ctypes.data, 0)
    expect_failure(handle.cc_gather_dense, m.device_handle(), None, 0, 4, g + 1, dense.ctypes.data, 0)   # ld not a multiple of 4
    # malformed CSR: unsorted columns in one row (found by the device-side check after the upload)
    indptr = np.array([0, 3, 5], dtype=np.int64)
    idx = np.array([0, 5, 2, 1, 7], dtype=np.int32)
    vals = np.ones(5, dtype=np.float32)
    h = ctypes.c_void_p()
    expect_failure(handle.cc_csr_upload, indptr.ctypes.data, idx.ctypes.data, vals.ctypes.data, 2, 5, 16, ctypes.byref(h))
    idx2 = np.array([0, 2, 99, 1, 7], dtype=np.int32)
    expect_failure(handle.cc_csr_upload, indptr.ctypes.data, idx2.ctypes.data, vals.ctypes.data, 2, 5, 16, ctypes.byref(h))
    # kNN with k larger than the candidate set, after the upload of x
    x = np.random.rand(10, 8).astype(np.float32)
    oi = np.empty((10, 15), dtype=np.int32)
    od = np.empty((10, 15), dtype=np.float32)
    expect_failure(handle.cc_knn_l2, x.ctypes.data, 10, 8, 15, 0, oi.ctypes.data, od.ctypes.data)
    # encoder of another gene count against this matrix
    other = cc.HipEncoder(8, g + 32, 'cls', seed=i)
    expect_failure(handle.cc_encode, other._h, m.device_handle(), None, 0, 4, out.ctypes.data)
    del other
    # NULL handles / NULL buffers: refused with a message, nothing dereferenced
    r, c = ctypes.c_int64(), ctypes.c_int64()
    expect_failure(handle.cc_encode, None, m.device_handle(), None, 0, 4, out.ctypes.data)
    expect_failure(handle.cc_encode, enc._h, None, None, 0, 4, out.ctypes.data)
    expect_failure(handle.cc_encode, enc._h, m.device_handle(), None, 0, 4, None)
    expect_failure(handle.cc_encode_dense, None, dense.ctypes.data, 4, out.ctypes.data)
    expect_failure(handle.cc_gather_dense, None, None, 0, 4, g + 8, dense.ctypes.data, 0)
    expect_failure(handle.cc_encoder_weight_shape, None, 0, ctypes.byref(r), ctypes.byref(c))
    expect_failure(handle.cc_encoder_set_low_latency, None, 1)
    expect_failure(handle.cc_encoder_set_sparse_input, None, 1)
    expect_failure(handle.cc_encoder_set_batch, None, 64)
    expect_failure(handle.cc_generate_dense, None, dense.ctypes.data, 4, dense.ctypes.data)
    assert handle.cc_encoder_num_weights(None) == -1 and handle.cc_generator_num_weights(None) == -1
    assert handle.cc_encoder_fp32_fallback_passes(None) == -1 and handle.cc_encoder_batch(None) == -1
    expect_failure(handle.cc_knn_l2, None, 10, 8, 3, 0, oi.ctypes.data, od.ctypes.data)


def cycle(i):
    m = synth.synth_csr(300 + i % 7, 2000, density=0.05, seed=i)
    for dtype in ('f32', 'f16x2', 'f16'):
        enc = cc.HipEncoder(16, 2000, 'cont' if i % 2 else 'cls', seed=i, dtype=dtype, batch_rows=128)
        z = enc.predict(m)
        enc.predict(m.to_dense()[:200])                 # the dense-host pipeline (two slots, two streams)
        res = pipeline.encode_and_knn(enc, m, 5)
        if dtype == 'f32':
            failing_calls(i, m, enc)
        del enc, res
    cc.knn_graph(z, 7)
    ws = ctypes.c_void_p()
    _lib.check(handle.cc_knn_create(ctypes.byref(ws)))  # a workspace handle of its own
    zt = torch.from_numpy(z).cuda()
    oi = np.empty((len(z), 7), dtype=np.int32)
    od = np.empty((len(z), 7), dtype=np.float32)
    _lib.check(handle.cc_knn_search_dev(ws, zt.data_ptr(), len(z), z.shape[1], 0, len(z), 7, 0, oi.ctypes.data, od.ctypes.data))
    handle.cc_knn_free(ws)
    gen = build_cont_generator(16, 2000)
    gen.predict((np.random.rand(8, 16).astype(np.float32), np.random.rand(8, 16).astype(np.float32)))
    del gen, m, zt
    gc.collect()


start = free_mb()
for i in range(3):
    cycle(i)                       # warm the grow-only scratch buffers
base = free_mb()
for i in range(3, 63):
    cycle(i)
    if i % 20 == 2:
        print(f'after {i - 2} cycles: free {free_mb():.1f} MiB (start {base:.1f})', flush=True)
end = free_mb()
print('delta MiB over 60 cycles (with failing calls):', round(base - end, 1))
gc.collect()
_lib.shutdown()
after = free_mb()
print(f'after cc_shutdown: {after - end:+.1f} MiB returned by the context; {start - after:.1f} MiB below the level before the first call')
_lib.ensure_init(0)                # the library comes back after a shutdown
cycle(99)
print('re-initialised and ran one more cycle')
sys.exit(1 if base - end > 64 else 0)
