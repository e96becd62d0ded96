#!/usr/bin/env python3
"""Timings of the §8(f) pieces next to the hot path (one MI355X): generator forward, duplicate grouping, .mtx ingest.
Prints one line per piece; numbers quoted in DESIGN.md §1 come from here."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cellcomm_amd as cc
from cellcomm_amd import _lib, schema, synth
from cellcomm_amd.generator import HipGenerator

_lib.ensure_init(0)

# 1. generator forward (reference src/bigan_basic.py:39-44 -> src/bigan_cont.py:7-25), host in / host out
g, e, n = 30000, 64, 8192
gen = HipGenerator(e, g, 'cont', seed=3)
rng = np.random.default_rng(0)
zin = (rng.random((n, e), dtype=np.float32), rng.random((n, e), dtype=np.float32))
gen.predict(zin)
_lib.profile_enable(True); _lib.profile_reset()
t0 = time.perf_counter()
for _ in range(3):
    out = gen.predict(zin)
dt = (time.perf_counter() - t0) / 3
prof = _lib.profile_snapshot(); _lib.profile_enable(False)
kern_ms = sum(v['ms'] for v in prof.values()) / 3
mac = 2 * e * 50 + (50 + 2 * e) * 256 + 256 * 256 + (256 + 2 * e) * int(g * 0.1) + int(g * 0.1) * int(g * 0.2) + int(g * 0.2) * g
print(f'generator cont G={g} E={e}: {n} cells in {dt * 1e3:.1f} ms host to host ({n / dt:,.0f} cells/s; output {out.nbytes / 1e6:.0f} MB), '
      f'kernels {kern_ms:.1f} ms = {2 * mac * n / (kern_ms * 1e-3) / 1e12:.1f} TFLOP/s on {2 * mac / 1e6:.0f} MFLOP/cell')

# 2. duplicate-coordinate grouping (reference src/intercepts/db_recorder.py:113-117)
for rows in (100000, 1000000):
    xyz = rng.random((rows, 3), dtype=np.float32)
    xyz[rng.choice(rows, rows // 20, replace=False)] = xyz[rng.choice(rows, rows // 20)]
    schema.find_duplicate_ids_gpu(xyz)
    t0 = time.perf_counter()
    groups = schema.find_duplicate_ids_gpu(xyz)
    print(f'duplicate groups of {rows} rows: {(time.perf_counter() - t0) * 1e3:.1f} ms ({len(groups)} groups)')

# 3. .mtx ingest (reference src/cell_type_training.py:9-17), host only
cells, genes = 20000, 30000
m = synth.synth_csr(cells, genes, seed=5)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, 'matrix.mtx')
    rows = np.repeat(np.arange(cells), np.diff(m.indptr))
    with open(path, 'w') as f:
        f.write('%%MatrixMarket matrix coordinate integer general\n%\n')
        f.write(f"{genes} {cells} {len(m.indices)}\n")
        np.savetxt(f, np.stack([m.indices + 1, rows + 1, m.values.astype(np.int64)], axis=1), fmt='%d')
    size = os.path.getsize(path)
    t0 = time.perf_counter()
    lm = cc.load_matrix(path)
    dt = time.perf_counter() - t0
    print(f'.mtx ingest: {len(m.indices):,} triplets ({size / 1e6:.0f} MB) in {dt:.2f} s = {len(m.indices) / dt / 1e6:.1f} M triplets/s, shape {lm.shape}')
