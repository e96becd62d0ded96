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


def read_enc_file(path):
    """Reads a file written by write_enc_file (never a reference-shipped pickle)."""
    with open(path, 'rb') as f:
        return np.asarray(pickle.load(f))


def load_barcodes(barcode_file):
    with open(barcode_file) as f:
        return [line.strip() for line in f.readlines()]
