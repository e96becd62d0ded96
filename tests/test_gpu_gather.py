"""K1 CSR->dense gather on the GPU: bit-exact against the host densification."""
import ctypes

import numpy as np
import pytest

from cellcomm_amd import _lib, matrix, synth

pytestmark = pytest.mark.gpu


def gather(m, ld, row_ids=None, row_start=0, n=None):
    lib = _lib.ensure_init(0)
    n = (len(row_ids) if row_ids is not None else m.shape[0] - row_start) if n is None else n
    out = np.full((n, ld), np.nan, dtype=np.float32)
    ids = None if row_ids is None else np.ascontiguousarray(row_ids, dtype=np.int64)
    _lib.check(lib.cc_gather_dense(m.device_handle(), None if ids is None else ids.ctypes.data, row_start, n, ld,
                                   out.ctypes.data, 0))
    return out


@pytest.mark.parametrize('n,g,density', [(37, 100, 0.2), (64, 5000, 0.05), (16, 30000, 0.05), (5, 4097, 0.5), (9, 3, 0.9)])
def test_gather_matches_dense(hip, n, g, density):
    m = synth.synth_csr(n, g, density=density, seed=n + g)
    ld = (g + 31) // 32 * 32
    out = gather(m, ld)
    assert np.array_equal(out[:, :g], m.to_dense())
    assert np.all(out[:, g:] == 0)                      # the padding is written as zeros, not left stale


def test_gather_row_ids_and_ranges(hip):
    m = synth.synth_csr(200, 9000, seed=5)
    dense = m.to_dense()
    ids = np.random.default_rng(0).choice(200, 77, replace=False)
    assert np.array_equal(gather(m, 9024, row_ids=ids)[:, :9000], dense[ids])
    assert np.array_equal(gather(m, 9000, row_start=150, n=50), dense[150:])
    rep = np.array([3, 3, 3, 199, 0])                   # repeated rows are legal
    assert np.array_equal(gather(m, 9000, row_ids=rep), dense[rep])


def test_gather_empty_rows_and_example(hip, golden_dir, known):
    m = matrix.CellMatrix(np.array([0, 0, 2, 2, 2]), np.array([1, 6]), np.array([4.0, 9.0]), 8)
    out = gather(m, 8)
    want = np.zeros((4, 8), dtype=np.float32)
    want[1, 1], want[1, 6] = 4, 9
    assert np.array_equal(out, want)
    ex = matrix.load_matrix(f'{golden_dir}/example_matrix.mtx')
    assert np.array_equal(gather(ex, 32)[:, :5], np.array(known['load_matrix']['matrix'], dtype=np.float32))


def test_gather_rejects_bad_arguments(hip):
    m = synth.synth_csr(10, 50, density=0.3, seed=1)
    with pytest.raises(ValueError):
        gather(m, 52, row_ids=np.array([0, 10]))        # row id out of range
    with pytest.raises(ValueError):
        gather(m, 48)                                   # ld < G
    with pytest.raises(ValueError):
        gather(m, 52, row_start=5, n=6)
    with pytest.raises(ValueError):                     # unsorted columns are refused at upload
        matrix.CellMatrix(np.array([0, 2]), np.array([5, 1]), np.array([1.0, 1.0]), 8).device_handle()


def test_gather_to_device_pointer(hip):
    import torch
    m = synth.synth_csr(33, 700, density=0.1, seed=2)
    out = torch.full((33, 704), float('nan'), device='cuda')
    torch.cuda.synchronize()
    _lib.check(hip.cc_gather_dense(m.device_handle(), None, 0, 33, 704, ctypes.c_void_p(out.data_ptr()), 1))
    assert np.array_equal(out.cpu().numpy()[:, :700], m.to_dense())
