"""Synthetic 10x-style inputs: per cell nnz ~ Binomial(G, density),
distinct sorted gene ids, integer counts 1 + Geometric(0.5) clipped to [1, 2047]
(exact in float32/float16), every gene present at least once."""
import numpy as np

from .matrix import CellMatrix


def synth_csr(n_cells, n_genes, density=0.05, seed=1234):
    """Host (NumPy) generator — deterministic; for tests and small runs."""
    rng = np.random.default_rng(seed)
    counts = rng.binomial(n_genes, density, size=n_cells)
    indptr = np.zeros(n_cells + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = np.empty(indptr[-1], dtype=np.int32)
    for r in range(n_cells):
        indices[indptr[r]:indptr[r + 1]] = np.sort(rng.choice(n_genes, size=counts[r], replace=False))
    values = np.minimum(rng.geometric(0.5, size=indptr[-1]), 2047).astype(np.float32)
    m = CellMatrix(indptr, indices, values, n_genes)
    return _cover_all_genes(m, rng)


def _cover_all_genes(m, rng):
    seen = np.zeros(m.shape[1], dtype=bool)
    seen[m.indices] = True
    missing = np.nonzero(~seen)[0]
    if len(missing) == 0 or m.shape[0] == 0:
        return m
    dense_rows = {}
    for g in missing:                      # one extra entry per missing gene, spread over cells
        dense_rows.setdefault(int(rng.integers(m.shape[0])), []).append(int(g))
    indptr, indices, values = [0], [], []
    for r in range(m.shape[0]):
        idx = m.indices[m.indptr[r]:m.indptr[r + 1]]
        val = m.values[m.indptr[r]:m.indptr[r + 1]]
        if r in dense_rows:
            extra = np.array(dense_rows[r], dtype=np.int32)
            idx = np.concatenate([idx, extra])
            val = np.concatenate([val, np.ones(len(extra), dtype=np.float32)])
            order = np.argsort(idx, kind='stable')
            idx, val = idx[order], val[order]
        indices.append(idx)
        values.append(val)
        indptr.append(indptr[-1] + len(idx))
    return CellMatrix(np.array(indptr), np.concatenate(indices), np.concatenate(values), m.shape[1])


def synth_csr_device(n_cells, n_genes, density=0.05, seed=1234, device='cuda', rows_per_chunk=2048):
    """Same distribution generated with torch directly in HBM (bench-sized inputs:
    100k x 30k has 150M non-zeros).  Returns (CellMatrix over device arrays,
    (indptr, indices, values) torch tensors)."""
    import torch
    from .matrix import wrap_device_csr
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    idx_chunks, val_chunks, count_chunks = [], [], []
    for r0 in range(0, n_cells, rows_per_chunk):
        rows = min(rows_per_chunk, n_cells - r0)
        mask = torch.rand((rows, n_genes), device=device, generator=gen) < density
        nz = mask.nonzero(as_tuple=False)                         # row-major => sorted columns per row
        count_chunks.append(mask.sum(dim=1))
        idx_chunks.append(nz[:, 1].to(torch.int32))
        u = torch.rand(nz.shape[0], device=device, generator=gen).clamp_min(1e-12)
        geo = torch.floor(torch.log(u) / np.log(0.5)).to(torch.float32) + 1.0   # Geometric(0.5) on {1,2,..}
        val_chunks.append(geo.clamp_(1.0, 2047.0))
        del mask, nz, u, geo
    counts = torch.cat(count_chunks)
    indptr = torch.zeros(n_cells + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, dim=0)
    indices = torch.cat(idx_chunks).contiguous()
    values = torch.cat(val_chunks).contiguous()
    torch.cuda.synchronize()
    nnz = int(indptr[-1].item())
    m = wrap_device_csr(indptr.data_ptr(), indices.data_ptr(), values.data_ptr(), n_cells, nnz, n_genes,
                        keepalive=(indptr, indices, values))
    return m, (indptr, indices, values)
