"""Cell x gene matrix kept as CSR (host side) — the data layer of the hot path.

Mirrors reference src/cell_type_training.py:9-26.  ``load_matrix`` has the
pivot semantics of the reference (rows = sorted distinct barcodes present,
columns = sorted distinct genes present, duplicate (barcode, gene) entries
averaged, absent entries 0, the three header lines skipped and the declared
dimensions ignored) but never builds the dense N x G table: at 100k x 30k that
table is 24 GB of float64, which is why the reference cannot run the large
configs at all.
"""
import ctypes

import numpy as np
import pandas as pd

from . import _lib


class CellMatrix:
    """CSR cell x gene matrix with the few DataFrame members the hot path uses
    (``shape``, ``index``, ``columns``, ``sample``, ``__len__``)."""

    def __init__(self, indptr, indices, values, n_genes, index=None, columns=None):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.values = np.ascontiguousarray(values, dtype=np.float32)
        n_rows = len(self.indptr) - 1
        self.shape = (n_rows, int(n_genes))
        self.index = np.arange(n_rows) if index is None else np.asarray(index)
        self.columns = np.arange(n_genes) if columns is None else np.asarray(columns)
        self._row_ids = None      # set on sampled views: rows of the parent, in sampled order
        self._parent = None
        self._device = None       # opaque cc_csr handle

    # -- construction -----------------------------------------------------------
    @classmethod
    def from_dense(cls, dense, index=None, columns=None):
        dense = np.asarray(dense, dtype=np.float32)
        rows, cols = np.nonzero(dense)
        indptr = np.zeros(dense.shape[0] + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=dense.shape[0]), out=indptr[1:])
        return cls(indptr, cols.astype(np.int32), dense[rows, cols], dense.shape[1], index, columns)

    @classmethod
    def from_scipy(cls, sp):
        sp = sp.tocsr().copy()
        sp.sum_duplicates()
        sp.sort_indices()
        return cls(sp.indptr, sp.indices, sp.data, sp.shape[1])

    # -- DataFrame-like surface --------------------------------------------------
    def __len__(self):
        return self.shape[0]

    @property
    def nnz(self):
        if self.indptr is None:                       # device-only matrix (wrap_device_csr)
            return self._nnz
        if self._row_ids is None:
            return int(self.indptr[-1])
        return int((self._parent.indptr[self._row_ids + 1] - self._parent.indptr[self._row_ids]).sum())

    def sample(self, n, random_state=None):
        """Rows drawn without replacement, like DataFrame.sample (reference
        src/cell_type_training.py:37-38): positions are
        RandomState(seed).choice(len, n, replace=False), exactly what pandas does.  With
        random_state=None pandas draws from NumPy's GLOBAL state (pandas.core.common.random_state
        returns np.random for None), so a caller's np.random.seed(...) reproduces the batches of
        CellTraining.run() here exactly as it does with the reference."""
        if random_state is None:
            rs = np.random.mtrand._rand
        elif isinstance(random_state, np.random.RandomState):
            rs = random_state
        else:
            rs = np.random.RandomState(random_state)
        locs = rs.choice(self.shape[0], size=n, replace=False)
        return self.take(locs)

    def take(self, locs):
        """View of the given row positions (no copy of the CSR arrays)."""
        locs = np.asarray(locs, dtype=np.int64)
        base = self if self._row_ids is None else self._parent
        rows = locs if self._row_ids is None else self._row_ids[locs]
        view = CellMatrix.__new__(CellMatrix)
        view.indptr, view.indices, view.values = base.indptr, base.indices, base.values
        view.shape = (len(rows), base.shape[1])
        view.index = base.index[rows]
        view.columns = base.columns
        view._row_ids = rows
        view._parent = base
        view._device = None
        return view

    def to_dense(self, dtype=np.float32):
        """Host densification (for small matrices / interop; not a compute path)."""
        base = self if self._row_ids is None else self._parent
        rows = np.arange(base.shape[0]) if self._row_ids is None else self._row_ids
        out = np.zeros((len(rows), base.shape[1]), dtype=dtype)
        for o, r in enumerate(rows):
            lo, hi = base.indptr[r], base.indptr[r + 1]
            out[o, base.indices[lo:hi]] = base.values[lo:hi]
        return out

    def to_dataframe(self):
        return pd.DataFrame(self.to_dense(np.float64), index=pd.Index(self.index, name='barcode'),
                            columns=pd.Index(self.columns, name='gene'))

    def __array__(self, dtype=None, copy=None):
        return self.to_dense(dtype or np.float32)

    # -- device residency ----------------------------------------------------------
    def device_handle(self):
        """cc_csr handle of the (parent) matrix, uploaded on first use."""
        base = self if self._row_ids is None else self._parent
        if base._device is None:
            handle = _lib.ensure_init()
            h = ctypes.c_void_p()
            _lib.check(handle.cc_csr_upload(base.indptr.ctypes.data, base.indices.ctypes.data,
                                            base.values.ctypes.data, base.shape[0], int(base.indptr[-1]),
                                            base.shape[1], ctypes.byref(h)))
            base._device = _DeviceCsr(h)
        return base._device.handle

    def row_ids(self):
        return self._row_ids


class _DeviceCsr:
    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().cc_csr_free(self.handle)
        except Exception:
            pass


def wrap_device_csr(dev_indptr, dev_indices, dev_vals, n_rows, nnz, n_genes, keepalive=None):
    """CellMatrix over CSR arrays that already live in HBM (e.g. torch tensors);
    the arrays are adopted, not copied.  `keepalive` is held to pin them."""
    handle = _lib.ensure_init()
    h = ctypes.c_void_p()
    _lib.check(handle.cc_csr_wrap_device(dev_indptr, dev_indices, dev_vals, n_rows, nnz, n_genes, ctypes.byref(h)))
    m = CellMatrix.__new__(CellMatrix)
    m.indptr = m.indices = m.values = None
    m.shape = (int(n_rows), int(n_genes))
    m.index = np.arange(n_rows)
    m.columns = np.arange(n_genes)
    m._row_ids = None
    m._parent = None
    m._device = _DeviceCsr(h)
    m._keepalive = keepalive
    m._nnz = int(nnz)
    return m


def load_matrix(matrix_file, verbose=False, engine='native'):
    """10x-style MatrixMarket triplets -> CellMatrix.  Same result as the pivot at
    reference src/cell_type_training.py:9-17 (see module docstring).  engine='native' streams the
    file through the C++ parser in libcellcomm_hip.so (cc_mtx_load, host only); engine='pandas'
    is the NumPy/pandas statement of the same semantics."""
    if verbose:
        print(f'============ Loading {matrix_file}...')
    if engine == 'native':
        m = _load_matrix_native(matrix_file)
        if verbose:
            print(f'============ DONE! barcodes: {m.shape[0]}, genes: {m.shape[1]}')
        return m
    if engine != 'pandas':
        raise ValueError(f'unknown engine {engine!r}')
    df = pd.read_csv(matrix_file, header=None, skiprows=3, sep=r'\s+', names=['gene', 'barcode', 'p'])
    genes, gene_pos = np.unique(df['gene'].to_numpy(), return_inverse=True)
    barcodes, row_pos = np.unique(df['barcode'].to_numpy(), return_inverse=True)
    p = df['p'].to_numpy(dtype=np.float64)
    n_genes = len(genes)
    key = row_pos.astype(np.int64) * max(n_genes, 1) + gene_pos.astype(np.int64)
    ukey, inv, counts = np.unique(key, return_inverse=True, return_counts=True)
    sums = np.bincount(inv, weights=p, minlength=len(ukey))
    vals = sums / counts                                  # pivot_table's default aggfunc='mean'
    rows = ukey // max(n_genes, 1)
    cols = ukey % max(n_genes, 1)
    indptr = np.zeros(len(barcodes) + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=len(barcodes)), out=indptr[1:])
    m = CellMatrix(indptr, cols.astype(np.int32), vals.astype(np.float32), n_genes, index=barcodes, columns=genes)
    if verbose:
        print(f'============ DONE! barcodes: {m.shape[0]}, genes: {m.shape[1]}')
    return m


def _load_matrix_native(matrix_file):
    import os
    lib = _lib.lib()
    h = ctypes.c_void_p()
    _lib.check(lib.cc_mtx_load(os.fsencode(os.fspath(matrix_file)), ctypes.byref(h)))
    try:
        n_rows, n_genes, nnz = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.cc_mtx_shape(h, ctypes.byref(n_rows), ctypes.byref(n_genes), ctypes.byref(nnz)))
        indptr = np.empty(n_rows.value + 1, dtype=np.int64)
        indices = np.empty(nnz.value, dtype=np.int32)
        vals = np.empty(nnz.value, dtype=np.float32)
        barcodes = np.empty(n_rows.value, dtype=np.int64)
        genes = np.empty(n_genes.value, dtype=np.int64)
        _lib.check(lib.cc_mtx_export(h, indptr.ctypes.data, indices.ctypes.data, vals.ctypes.data,
                                     barcodes.ctypes.data, genes.ctypes.data))
    finally:
        lib.cc_mtx_free(h)
    return CellMatrix(indptr, indices, vals, n_genes.value, index=barcodes, columns=genes)


def load_cells(cells_file, verbose=False):
    """Dense CSV written by the reference's `convert` command (reference
    src/cell_type_training.py:20-26) -> CellMatrix."""
    if verbose:
        print(f'============ Loading {cells_file}...')
    df = pd.read_csv(cells_file, index_col=0)
    m = CellMatrix.from_dense(df.to_numpy(), index=df.index.to_numpy(), columns=df.columns.to_numpy())
    if verbose:
        print(f'============ DONE! barcodes: {m.shape[0]}, genes: {m.shape[1]}')
    return m


def store_cells(matrix, cell_file, block_rows=256):
    """Write a CellMatrix as the dense CSV `DataFrame.to_csv` produces for the pivot of load_matrix
    (index column `barcode`, one column per gene, shortest round-trip decimals), streamed in row blocks
    so the dense table never exists in memory.  `load_cells` reads it back."""
    cols = np.asarray(matrix.columns if matrix.columns is not None else np.arange(matrix.shape[1])).tolist()
    idx = np.asarray(matrix.index if matrix.index is not None else np.arange(matrix.shape[0])).tolist()
    with open(cell_file, 'w', newline='') as f:
        f.write('barcode,' + ','.join(str(c) for c in cols) + '\n')
        for lo in range(0, matrix.shape[0], block_rows):
            hi = min(lo + block_rows, matrix.shape[0])
            # values are held as float32: their shortest round-trip decimals are what pandas prints for the
            # float64 originals whenever those had <= 7 significant digits (counts and their means do)
            text = matrix.take(np.arange(lo, hi)).to_dense().astype(np.float32).astype(str)
            for r in range(hi - lo):
                f.write(str(idx[lo + r]))
                f.write(',')
                f.write(','.join(text[r].tolist()))
                f.write('\n')


def store_converted_cell_file(matrix_file, cell_file):
    """The reference's `convert` step (reference src/__main__.py:76-81): .mtx triplets -> dense cell CSV."""
    print('converting matrix file:')
    m = load_matrix(matrix_file, verbose=True)
    print('storing cell file:', cell_file, '... ', end='', flush=True)
    store_cells(m, cell_file)
    print('done')
