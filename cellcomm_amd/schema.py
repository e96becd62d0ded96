"""Output schema the cellan viewer consumes — built without Mongo.

* ``.enc`` files and the legacy ``encs`` document: reference
  import_encodings.py:24-63 (``{_id, pids, ns, xs, ys, zs}``, coords = enc x 255).
* ``encits`` document: reference src/intercepts/db_recorder.py:82-121 /
  db_schema.js:14-26 (``{eid, it, cids, ns, xs, ys, zs, ds}``).
* run-level ``encs`` document ``{_id, date, defit, showits, srcs}``: reference
  src/intercepts/db_recorder.py:48-64 (the first thing cellan fetches,
  cellan/backend/apiRouter.js:8-15); `recorder.DbRecorder` keeps it up to date.
* optional additive fields ``nn`` / ``nd`` carry the kNN graph; cellan ignores
  unknown fields (reference cellan/backend/apiRouter.js:17-25).
"""
import os
import pickle
from collections import defaultdict
from datetime import datetime

import numpy as np
import pandas as pd


def cell_id_from(ix):
    return ix + 1


def find_duplicate_ids(np_coords):
    """Groups (len > 1) of 1-based ids that share identical (x, y, z), in the group
    order of the reference (iteration order of a Python set of the coordinate
    tuples, src/intercepts/db_recorder.py:113-117) — in O(N) instead of O(N·U)."""
    coords = [(c[0], c[1], c[2]) for c in np_coords]
    members = defaultdict(list)
    for i, c in enumerate(coords):
        members[c].append(cell_id_from(i))
    return [members[uc] for uc in set(coords) if len(members[uc]) > 1]


def find_duplicate_ids_gpu(np_coords):
    """Same result as find_duplicate_ids with the grouping done on the GPU
    (cc_duplicate_representatives): only the UNIQUE coordinates pass through Python, to reproduce
    the reference's group order (iteration order of the set of coordinate tuples — inserting the
    distinct tuples in first-occurrence order builds the same set as inserting all of them; with NaN
    coordinates that order is not reproducible even by the reference itself, CPython hashes a NaN by
    object identity)."""
    import ctypes  # noqa: F401
    from . import _lib
    coords = np.ascontiguousarray(np_coords, dtype=np.float32)
    n = len(coords)
    if n == 0:
        return []
    rep = np.empty(n, dtype=np.int32)
    _lib.check(_lib.ensure_init().cc_duplicate_representatives(coords.ctypes.data, n, rep.ctypes.data))
    firsts = np.nonzero(rep == np.arange(n))[0]                 # one row per distinct coordinate (or NaN row)
    order = np.argsort(rep, kind='stable')
    bounds = np.searchsorted(rep[order], firsts)
    members = {int(f): (order[b:e] + 1).tolist()
               for f, b, e in zip(firsts, bounds, np.append(bounds[1:], n)) if e - b > 1}
    uniq = [(c[0], c[1], c[2]) for c in np.asarray(np_coords)[firsts]]
    first_of = {}
    for f, t in zip(firsts, uniq):
        first_of.setdefault(t, int(f))
    return [members[first_of[uc]] for uc in set(uniq) if first_of.get(uc) in members]


def encits_document(enc_run_id, it, encodings, barcodes, knn=None):
    encodings = np.asarray(encodings)
    enc_shape = encodings.shape
    assert enc_shape[0] == len(barcodes), \
        f'encodings + barcodes have different length: {enc_shape[0]} != {len(barcodes)}'
    assert enc_shape[1] == 3, \
        f'encodings vector length = {enc_shape[1]}, not in x, y, z format'
    coords = np.multiply(encodings, 255)
    doc = {
        'eid': enc_run_id,
        'it': it,
        'cids': [cell_id_from(i) for i in range(len(barcodes))],
        'ns': list(barcodes),
        'xs': coords[:, 0].tolist(),
        'ys': coords[:, 1].tolist(),
        'zs': coords[:, 2].tolist(),
        'ds': find_duplicate_ids(coords),
    }
    if knn is not None:
        idx, dist = knn
        doc['nn'] = (np.asarray(idx) + 1).tolist()       # 1-based like cids
        doc['nd'] = np.asarray(dist).tolist()
    return doc


def encits_from_wide(enc_run_id, it, encodings, barcodes, dims=(0, 1, 2), knn=None):
    """`encits` document of encodings wider than 3 (E = 32 / 64 in BASELINE configs[1..4]): the viewer plots exactly
    three coordinates (reference db_recorder.py:89-90 asserts width 3), so three latent dimensions are exported —
    by default dims 0..2, as import_encodings.py:44-54 does for the legacy document — while `knn` (computed on the
    full-width encodings) rides along unchanged."""
    encodings = np.asarray(encodings)
    assert encodings.ndim == 2 and len(dims) == 3, 'dims must name three latent dimensions'
    assert all(0 <= d < encodings.shape[1] for d in dims), \
        f'dims {tuple(dims)} outside encodings of width {encodings.shape[1]}'
    return encits_document(enc_run_id, it, encodings[:, list(dims)], barcodes, knn=knn)


def encs_run_document(enc_run_id, sources, now=None):
    """The run-level document of the `encs` collection (reference db_recorder.py:48-64, pinned by
    test/db_recorder_test.py:41-52): created once per run with `defit` 0 and no `showits`; the recorder's
    interceptor then sets `defit` to the latest iteration and appends to `showits` (db_recorder.py:104-108).
    `sources` = {'matrix', 'barcodes', 'genes'} paths; only their file names are stored."""
    def file_name(full_path):
        return full_path.split('/')[-1]
    return {
        '_id': enc_run_id,
        'date': now if now is not None else datetime.now(),
        'defit': 0,
        'showits': [],
        'srcs': {
            'matrix': file_name(sources['matrix']),
            'barcodes': file_name(sources['barcodes']),
            'genes': file_name(sources['genes']),
        },
    }


def legacy_encs_document(iteration, encodings, barcodes):
    coords = np.multiply(np.asarray(encodings), 255)
    if len(barcodes) != len(coords):
        raise ValueError(f'Coordinates-/barcode-data different length: {len(barcodes)} != {len(coords)}')
    return {
        '_id': iteration,
        'pids': [i + 1 for i in range(len(barcodes))],
        'ns': list(barcodes),
        'xs': coords[:, 0].tolist(),
        'ys': coords[:, 1].tolist(),
        'zs': coords[:, 2].tolist(),
    }


def legacy_iteration_id(enc_file_name):
    """Iteration id of an `.enc` file name as reference import_encodings.py:68 derives it:
    `enc_file.rstrip('.enc')` — which strips trailing CHARACTERS out of {'.', 'e', 'n', 'c'}, not the suffix:
    '5009.enc' -> '5009', but '100e.enc' -> '100' and 'dec.enc' -> 'd'."""
    return enc_file_name.rstrip('.enc')


def legacy_import_directory(encodings_dir, barcodes, collection=None, listdir=os.listdir):
    """The legacy importer's main loop (reference import_encodings.py:66-77): one `encs` document per entry of
    `logs/<run>/encodings`, in directory-listing order, `_id` = `legacy_iteration_id(name)`, coordinates read
    from `<_id>.enc` — as the reference's `load_coords(iteration)` does, so a name the id rule mangles
    ('100e.enc') loads '100.enc' (or raises FileNotFoundError when that does not exist).  `barcodes`: list of
    names or the path of a barcodes.tsv.  Documents are returned and, when `collection` is given, passed to its
    `insert_one` one by one (a pymongo collection or `recorder.MemoryDatabase()[...]`)."""
    if isinstance(barcodes, (str, os.PathLike)):
        barcodes = load_barcodes(barcodes)
    docs = []
    for enc_file in listdir(encodings_dir):
        iteration = legacy_iteration_id(enc_file)
        doc = legacy_encs_document(iteration, read_enc_file(os.path.join(encodings_dir, f'{iteration}.enc')), barcodes)
        if collection is not None:
            collection.insert_one(doc)
        docs.append(doc)
    return docs


def write_enc_file(path, encodings):
    """``logs/<run>/encodings/<it>.enc`` = pickle of the N x E array (the format
    reference import_encodings.py:24-27 reads back)."""
    with open(path, 'wb') as f:
        pickle.dump(np.asarray(encodings), f)


class _ArrayOnlyUnpickler(pickle.Unpickler):
    """`.enc` files are pickles (the format reference import_encodings.py:24-27 reads).  Loading one
    must not execute anything: only the globals needed to rebuild a plain ndarray / nested lists of
    numbers are resolvable, everything else is refused."""
    _ALLOWED = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
                ('numpy', 'ndarray'), ('numpy', 'dtype'), ('numpy.core.multiarray', 'scalar'),
                ('numpy._core.multiarray', 'scalar')}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'.enc file references {module}.{name}: only plain arrays are accepted')


def read_enc_file(path):
    """Encodings array of an `.enc` file (array-only unpickling, see _ArrayOnlyUnpickler)."""
    with open(path, 'rb') as f:
        return np.asarray(_ArrayOnlyUnpickler(f).load())


def load_barcodes(barcode_file):
    with open(barcode_file) as f:
        return [line.strip() for line in f.readlines()]


def load_genes(genes_file):
    """[(ensembl id, symbol), ...] of a 10x genes.tsv (reference src/intercepts/import_barcodes.py:78)."""
    with open(genes_file) as f:
        return [tuple(line.strip().split('\t')[:2]) for line in f.readlines()]


def cells_and_genes_documents(source_id, matrix_file, barcodes_file, genes_file):
    """The `cells` and `genes` collections the reference builds next to the encodings (reference
    src/intercepts/import_barcodes.py:14-45,74-81), without Mongo and without its per-triplet Python loop:

    * one cell document `{sid, cid, n, g: [{e, m, v}, ...]}` per RUN of consecutive triplets with the same
      barcode column (the reference compares the column text with the previous line's), `g` sorted by value,
      descending, ties in file order;
    * one gene document `{sid, e, m, cids}` per ensembl id, in order of first appearance, `cids` in file order.

    Returns (cells, genes).  The triplets are read once with pandas and grouped with NumPy; only the documents
    themselves are Python objects."""
    trip = pd.read_csv(matrix_file, header=None, skiprows=3, sep=' ', names=['gene', 'barcode', 'p'],
                       dtype={'gene': np.int64, 'barcode': str}, skipinitialspace=False)
    barcodes = load_barcodes(barcodes_file)
    genes_src = load_genes(genes_file)
    gene = trip['gene'].to_numpy()
    bc_text = trip['barcode'].to_numpy()
    cid = bc_text.astype(np.int64)
    val = trip['p'].to_numpy().astype(np.int64)                  # int(p_val)
    n = len(gene)
    if n == 0:
        return [], []
    new_run = np.ones(n, dtype=bool)
    new_run[1:] = bc_text[1:] != bc_text[:-1]
    run = np.cumsum(new_run) - 1
    # within a run: by value descending, stable
    order = np.lexsort((np.arange(n), -val, run))
    starts = np.flatnonzero(new_run)
    ends = np.append(starts[1:], n)
    e_of = [g[0] for g in genes_src]
    m_of = [g[1] for g in genes_src]
    g_sorted = gene[order] - 1
    v_sorted = val[order].tolist()
    cells = []
    for s, t in zip(starts.tolist(), ends.tolist()):
        c = int(cid[s])
        cells.append({'sid': source_id, 'cid': c, 'n': barcodes[c - 1],
                      'g': [{'e': e_of[g], 'm': m_of[g], 'v': v} for g, v in zip(g_sorted[s:t].tolist(), v_sorted[s:t])]})
    # genes: ensembl ids in order of first appearance; the reference keys by ensembl id, so two gene lines that
    # share one collapse into the first's document
    ens_code = {}
    codes = np.empty(len(genes_src), dtype=np.int64)
    for i, e in enumerate(e_of):
        codes[i] = ens_code.setdefault(e, len(ens_code))
    tcode = codes[gene - 1]
    first = np.full(len(ens_code), n, dtype=np.int64)
    np.minimum.at(first, tcode, np.arange(n))
    by_gene = np.argsort(tcode, kind='stable')
    counts = np.bincount(tcode, minlength=len(ens_code))
    offs = np.concatenate([[0], np.cumsum(counts)])
    cid_by_gene = cid[by_gene].tolist()
    first_line_of_code = {}
    for i, e in enumerate(e_of):
        first_line_of_code.setdefault(ens_code[e], i)
    genes = []
    for code in np.argsort(first, kind='stable').tolist():
        if counts[code] == 0:
            continue
        i = first_line_of_code[code]
        # `m` is the symbol of the gene line of the id's FIRST triplet (setdefault keeps the first document)
        line = int(gene[first[code]] - 1)
        genes.append({'sid': source_id, 'e': e_of[i], 'm': m_of[line], 'cids': cid_by_gene[offs[code]:offs[code + 1]]})
    return cells, genes
