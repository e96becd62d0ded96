"""Oracle: output documents consumed by cellan.  TEST INFRASTRUCTURE ONLY.

Restates reference import_encodings.py:24-63 (legacy ``encs`` document) and
reference src/intercepts/db_recorder.py:82-121 (``encits`` document and
find_duplicate_ids) and reference src/intercepts/import_barcodes.py:4-45,74-81
(``cells`` / ``genes`` documents) without Mongo.  Pinned by test/db_recorder_test.py:60-106,120-169
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


# ---- `cells` / `genes` documents (reference src/intercepts/import_barcodes.py:4-45,74-81), loop form ----
def load_file(file, converter, skip=0):
    with open(file) as f:
        return [converter(line) for line in f.readlines()[skip:]]


def convert_matrix(source_id, barcodes, genes_src, matrix):
    # import_barcodes.py:14-45: a new cell record whenever the barcode COLUMN TEXT differs from the previous
    # line's; genes keyed by ensembl id in order of first appearance; per-cell genes sorted by value, descending
    current_id, cells, genes, cell_record = 0, [], {}, {}
    for gene_line_num, barcode_line_num, p_val in matrix:
        cell_id = int(barcode_line_num)
        if barcode_line_num != current_id:
            current_id = barcode_line_num
            cell_record = {'sid': source_id, 'cid': cell_id, 'n': barcodes[cell_id - 1], 'g': []}
            cells.append(cell_record)
        ensembl, mgi = genes_src[int(gene_line_num) - 1][0], genes_src[int(gene_line_num) - 1][1]
        cell_record['g'].append({'e': ensembl, 'm': mgi, 'v': int(p_val)})
        genes.setdefault(ensembl, {'sid': source_id, 'e': ensembl, 'm': mgi, 'cids': []})['cids'].append(cell_id)
    for cell in cells:
        cell['g'].sort(key=lambda gene: gene['v'], reverse=True)
    return cells, [genes[k] for k in genes]


def cells_and_genes_documents(source_id, matrix_file, barcodes_file, genes_file):
    matrix = load_file(matrix_file, lambda line: line.strip().split(' '), skip=3)
    barcodes = load_file(barcodes_file, lambda line: line.strip())
    genes = load_file(genes_file, lambda line: line.strip().split('\t'))
    return convert_matrix(source_id, barcodes, genes, matrix)
