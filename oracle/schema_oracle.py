"""Oracle: output documents consumed by cellan.  TEST INFRASTRUCTURE ONLY.

Restates reference import_encodings.py:24-63 (legacy ``encs`` document) and
reference src/intercepts/db_recorder.py:82-121 (``encits`` document and
find_duplicate_ids) without Mongo.  Pinned by test/db_recorder_test.py:120-169
and the ValueError text at import_encodings.py:58.
"""
import numpy as np


def cell_id_from(ix):
    return ix + 1


def find_duplicate_ids(np_coords):
    # reference src/intercepts/db_recorder.py:113-117, verbatim semantics (O(N·U))
    coords = [(c[0], c[1], c[2]) for c in np_coords]
    unique_coords = set(coords)
    all_indices = [[cell_id_from(i) for i, x in enumerate(coords) if x == uc] for uc in unique_coords]
    return [ixs for ixs in all_indices if len(ixs) > 1]


def encits_document(enc_run_id, it, encodings, barcodes):
    enc_shape = encodings.shape
    assert enc_shape[0] == len(barcodes), \
        f'encodings + barcodes have different length: {enc_shape[0]} != {len(barcodes)}'
    assert enc_shape[1] == 3, \
        f'encodings vector length = {enc_shape[1]}, not in x, y, z format'
    coords = np.multiply(encodings, 255)
    return {
        'eid': enc_run_id, 'it': it,
        'cids': [cell_id_from(i) for i in range(len(barcodes))],
        'ns': list(barcodes),
        'xs': coords[:, 0].tolist(), 'ys': coords[:, 1].tolist(), 'zs': coords[:, 2].tolist(),
        'ds': find_duplicate_ids(coords),
    }


def legacy_encs_document(iteration, encodings, barcodes):
    coords = np.multiply(encodings, 255)
    if len(barcodes) != len(coords):
        raise ValueError(f'Coordinates-/barcode-data different length: {len(barcodes)} != {len(coords)}')
    return {
        '_id': iteration,
        'pids': [i + 1 for i in range(len(barcodes))],
        'ns': list(barcodes),
        'xs': coords[:, 0].tolist(), 'ys': coords[:, 1].tolist(), 'zs': coords[:, 2].tolist(),
    }
