"""Output schema the cellan viewer consumes — built without Mongo.

* ``.enc`` files and the legacy ``encs`` document: reference
  import_encodings.py:24-63 (``{_id, pids, ns, xs, ys, zs}``, coords = enc x 255).
* ``encits`` document: reference src/intercepts/db_recorder.py:82-121 /
  db_schema.js:14-26 (``{eid, it, cids, ns, xs, ys, zs, ds}``).
* optional additive fields ``nn`` / ``nd`` carry the kNN graph; cellan ignores
  unknown fields (reference cellan/backend/apiRouter.js:17-25).
"""
import pickle
from collections import defaultdict

import numpy as np


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
