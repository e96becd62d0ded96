"""Host-side mirror of the reference interface (no GPU): loader, sampling, schema
builders, BiGAN wrapper delegation, sharding arithmetic."""
import os
from unittest.mock import MagicMock

import numpy as np
import pytest

from conftest import GOLDEN
from cellcomm_amd import bigan, matrix, pipeline, schema, synth
from cellcomm_amd.encoder import _layer_plan, hidden_widths
from oracle import encoder_oracle as eo_mod
from oracle import loader_oracle as lo
from oracle import schema_oracle as so

MTX = os.path.join(GOLDEN, 'example_matrix.mtx')


def test_load_matrix_and_pivot(known):
    m = matrix.load_matrix(MTX)
    assert m.shape == (5, 5)
    assert np.array_equal(m.to_dense(), np.array(known['load_matrix']['matrix'], dtype=np.float32))
    assert m.index.tolist() == [1, 2, 3, 4, 5] and m.columns.tolist() == [1, 2, 3, 4, 5]


def _write_mtx(path, triplets):
    with open(path, 'w') as f:
        f.write('%%MatrixMarket matrix coordinate integer general\n%\n9 9 9\n')
        for g, b, p in triplets:
            f.write(f'{g} {b} {p}\n')


@pytest.mark.parametrize('engine', ['native', 'pandas'])
def test_load_matrix_engines_known_answer(known, engine):
    m = matrix.load_matrix(MTX, engine=engine)
    assert m.shape == (5, 5)
    assert np.array_equal(m.to_dense(), np.array(known['load_matrix']['matrix'], dtype=np.float32))


def test_native_parser_edge_cases(tmp_path):
    path = tmp_path / 'edge.mtx'
    with open(path, 'w') as f:                # CRLF, tabs, blank lines, float values, unordered, negative ids
        f.write('%%MatrixMarket\r\n%junk 1 2 3\r\n5 5 5\r\n')
        f.write('7\t3 2.5\r\n\n  2 3 1e1  \n7 3 3.5\n-1 9 4\n2 1 0\n')
    want = lo.load_matrix(path)
    for engine in ('native', 'pandas'):
        got = matrix.load_matrix(path, engine=engine)
        assert np.array_equal(got.index, want.index.to_numpy()) and np.array_equal(got.columns, want.columns.to_numpy())
        assert np.array_equal(got.to_dense(np.float64), want.to_numpy().astype(np.float64))
    bad = tmp_path / 'bad.mtx'
    bad.write_text('a\nb\nc\n1 2\n')
    with pytest.raises(ValueError):
        matrix.load_matrix(bad)
    with pytest.raises(ValueError):
        matrix.load_matrix(tmp_path / 'missing.mtx')
    empty = tmp_path / 'empty.mtx'
    empty.write_text('a\nb\nc\n')
    assert matrix.load_matrix(empty).shape == (0, 0)


def test_native_parser_large_file_matches_pandas_engine(tmp_path):
    rng = np.random.default_rng(8)
    n, g = 3000, 800
    m = synth.synth_csr(n, g, density=0.03, seed=4)
    path = tmp_path / 'big.mtx'
    rows = np.repeat(np.arange(n), np.diff(m.indptr))
    with open(path, 'w') as f:
        f.write('%%MatrixMarket matrix coordinate integer general\n%\n0 0 0\n')
        order = rng.permutation(len(rows)) if True else np.arange(len(rows))
        np.savetxt(f, np.stack([m.indices[order] + 1, rows[order] + 1, m.values[order].astype(np.int64)], 1), fmt='%d')
    a = matrix.load_matrix(path, engine='native')
    b = matrix.load_matrix(path, engine='pandas')
    for x, y in ((a.indptr, b.indptr), (a.indices, b.indices), (a.values, b.values), (a.index, b.index), (a.columns, b.columns)):
        assert np.array_equal(x, y)
    assert np.array_equal(a.indptr, m.indptr) and np.array_equal(a.indices, m.indices) and np.array_equal(a.values, m.values)


def test_load_matrix_equals_pivot_on_ragged_input(tmp_path):
    rng = np.random.default_rng(2)
    trip = [(int(rng.integers(1, 60)) * 3, int(rng.integers(1, 40)) * 2, int(rng.integers(0, 9))) for _ in range(900)]
    trip += trip[:50]                                   # exact duplicate entries -> averaged
    trip += [(5, 7, 3), (5, 7, 8)]                      # differing duplicates -> mean 5.5
    path = tmp_path / 'm.mtx'
    _write_mtx(path, trip)
    want = lo.load_matrix(path)
    got = matrix.load_matrix(path)
    assert got.shape == want.shape
    assert np.array_equal(got.index, want.index.to_numpy()) and np.array_equal(got.columns, want.columns.to_numpy())
    assert np.array_equal(got.to_dense(np.float64), want.to_numpy().astype(np.float32).astype(np.float64))
    assert np.all(np.diff(got.indptr) >= 0)
    for r in range(got.shape[0]):                       # canonical CSR: strictly increasing columns
        assert np.all(np.diff(got.indices[got.indptr[r]:got.indptr[r + 1]]) > 0)


def test_sample_cell_data_rows(known):
    ka = known['sample_cell_data']
    m = matrix.load_matrix(MTX)
    sampled = m.sample(ka['batch_size'], random_state=ka['seed'])
    assert sampled.shape == (3, 5)
    assert sampled.row_ids().tolist() == ka['rows']
    assert np.array_equal(sampled.to_dense(), np.array(known['load_matrix']['matrix'], dtype=np.float32)[ka['rows']])


@pytest.mark.parametrize('seed', [0, 1, 7, 123])
def test_sample_equals_pandas_sample(seed):
    m = synth.synth_csr(40, 30, density=0.2, seed=seed)
    df = m.to_dataframe()
    want = lo.sample_cell_data(df, 11, seed)
    got = m.sample(11, random_state=seed)
    assert np.array_equal(got.index, want.index.to_numpy())
    assert np.array_equal(got.to_dense(np.float64), want.to_numpy())
    again = got.take([3, 0])                            # view of a view
    assert np.array_equal(again.to_dense(np.float64), want.to_numpy()[[3, 0]])


def test_sample_without_seed_follows_numpy_global_state():
    """DataFrame.sample(n) with no seed draws from np.random's global state (what CellTraining.run() relies on,
    reference src/cell_type_training.py:37-38,42): seeding it must reproduce the same rows here."""
    m = synth.synth_csr(40, 30, density=0.2, seed=5)
    df = m.to_dataframe()
    for seed in (3, 11):
        np.random.seed(seed)
        want = [df.sample(7).index.to_numpy() for _ in range(3)]
        np.random.seed(seed)
        got = [m.sample(7).index for _ in range(3)]
        for w, g in zip(want, got):
            assert np.array_equal(g, w)


def test_load_cells_roundtrip(tmp_path):
    m = matrix.load_matrix(MTX)
    path = tmp_path / 'cells.csv'
    m.to_dataframe().to_csv(path)                       # what the reference's `convert` writes
    back = matrix.load_cells(path)
    assert np.array_equal(back.to_dense(), m.to_dense())


def test_store_converted_cell_file_writes_what_pandas_writes(tmp_path, capsys):
    # the reference's `convert` step: DataFrame.to_csv of the pivot (reference src/__main__.py:76-81)
    import pandas as pd
    mtx = MTX
    out = tmp_path / 'cells.csv'
    matrix.store_converted_cell_file(mtx, out)
    assert 'storing cell file:' in capsys.readouterr().out
    df = pd.read_csv(mtx, header=None, skiprows=3, sep=r'\s+', names=['gene', 'barcode', 'p'])
    want = df.pivot_table(index='barcode', columns='gene', values='p', fill_value=0).to_csv()
    assert out.read_text() == want
    back = matrix.load_cells(out)
    m = matrix.load_matrix(mtx)
    assert np.array_equal(back.to_dense(), m.to_dense())
    assert list(back.index) == list(m.index) and [int(c) for c in back.columns] == list(m.columns)
    # non-integer means (duplicate entries) keep their float64 repr
    ragged = tmp_path / 'dup.mtx'
    ragged.write_text('%%x\n%\n3 3 4\n1 1 1\n1 1 2\n3 2 7\n2 2 0.1\n')
    out2 = tmp_path / 'dup.csv'
    matrix.store_cells(matrix.load_matrix(ragged), out2)
    df2 = pd.read_csv(ragged, header=None, skiprows=3, sep=r'\s+', names=['gene', 'barcode', 'p'])
    assert out2.read_text() == df2.pivot_table(index='barcode', columns='gene', values='p', fill_value=0).to_csv()


def test_cells_and_genes_documents(tmp_path):
    # reference known answers (test/db_recorder_test.py:60-106) for its example files, then product == oracle
    # on files that exercise what the loop form does with repeated barcode runs, ties and shared ensembl ids
    src = 'example_barcodes.tsv'
    files = (MTX, os.path.join(GOLDEN, 'example_barcodes.tsv'), os.path.join(GOLDEN, 'example_genes.tsv'))
    for impl in (so.cells_and_genes_documents, schema.cells_and_genes_documents):
        cells, genes = impl(src, *files)
        assert len(cells) == 5 and len(genes) == 5
        assert cells[0] == {'sid': src, 'cid': 1, 'n': 'AAACCTGGTGTCCTCT-1', 'g': [
            {'e': 'ENSMUSG00000025902', 'm': 'Sox17', 'v': 11}, {'e': 'ENSMUSG00000102343', 'm': 'Gm37381', 'v': 6},
            {'e': 'ENSMUSG00000089699', 'm': 'Gm1992', 'v': 1}, {'e': 'ENSMUSG00000109048', 'm': 'Rp1', 'v': 1}]}
        assert cells[4] == {'sid': src, 'cid': 5, 'n': 'AAAGATGGTGATAAAC-1', 'g': [{'e': 'ENSMUSG00000109048', 'm': 'Rp1', 'v': 2}]}
        assert genes[0] == {'sid': src, 'e': 'ENSMUSG00000089699', 'm': 'Gm1992', 'cids': [1, 2, 3]}
        assert genes[4] == {'sid': src, 'e': 'ENSMUSG00000051951', 'm': 'Xkr4', 'cids': [2, 3, 4]}
    rng = np.random.default_rng(4)
    for case in range(6):
        nb, ng, nt = int(rng.integers(1, 30)), int(rng.integers(1, 25)), int(rng.integers(0, 300))
        bfile, gfile, mfile = tmp_path / f'b{case}.tsv', tmp_path / f'g{case}.tsv', tmp_path / f'm{case}.mtx'
        bfile.write_text(''.join(f'BC{i}-1\n' for i in range(nb)))
        ens = [f'ENS{int(rng.integers(0, max(1, ng - 2))):05d}' for _ in range(ng)]      # some ids shared by gene lines
        gfile.write_text(''.join(f'{e}\tSym{i}\n' for i, e in enumerate(ens)))
        bcs = rng.integers(1, nb + 1, nt)
        if case % 2 == 0:
            bcs = np.sort(bcs)                                                            # the usual 10x order
        lines = [f'{int(rng.integers(1, ng + 1))} {int(b)} {int(rng.integers(1, 6))}' for b in bcs]
        mfile.write_text('%%MatrixMarket\n%\n1 1 1\n' + ''.join(l + '\n' for l in lines))
        assert schema.cells_and_genes_documents('s', mfile, bfile, gfile) == so.cells_and_genes_documents('s', mfile, bfile, gfile)


def test_synth_csr_properties():
    m = synth.synth_csr(64, 200, density=0.05, seed=3)
    assert m.shape == (64, 200)
    assert set(np.unique(m.indices)) == set(range(200))            # every gene survives a pivot
    assert m.values.min() >= 1 and m.values.max() <= 2047 and np.all(m.values == np.round(m.values))
    for r in range(64):
        assert np.all(np.diff(m.indices[m.indptr[r]:m.indptr[r + 1]]) > 0)


# ---- schema ---------------------------------------------------------------------
def test_encits_document(known):
    ka = known['encits_document']
    doc = schema.encits_document('run', 2009, np.array(ka['encodings']), ka['ns'])
    assert doc == {'eid': 'run', 'it': 2009, **{k: ka[k] for k in ('cids', 'ns', 'xs', 'ys', 'zs', 'ds')}}
    assert schema.load_barcodes(os.path.join(GOLDEN, 'example_barcodes.tsv')) == ka['ns']


def test_schema_assertion_messages(known):
    msg = known['messages']
    with pytest.raises(AssertionError) as e:
        schema.encits_document('r', 1, np.array([[], [], [], []]), ['a'] * 5)
    assert str(e.value) == msg['length']
    with pytest.raises(AssertionError) as e:
        schema.encits_document('r', 1, np.array([[1, 2]] * 5), ['a'] * 5)
    assert str(e.value) == msg['xyz']
    with pytest.raises(ValueError) as e:
        schema.legacy_encs_document('1', np.zeros((4, 3)), ['a'] * 5)
    assert str(e.value) == msg['legacy_length']


def test_duplicate_groups_equal_reference_order():
    rng = np.random.default_rng(4)
    enc = rng.integers(0, 4, (300, 3)).astype(np.float32) / 4      # many exact duplicates
    coords = np.multiply(enc, 255)
    assert schema.find_duplicate_ids(coords) == so.find_duplicate_ids(coords)
    barcodes = [f'b{i}' for i in range(300)]
    assert schema.encits_document('e', 5, enc, barcodes) == so.encits_document('e', 5, enc, barcodes)
    assert schema.legacy_encs_document('7', enc, barcodes) == so.legacy_encs_document('7', enc, barcodes)


def test_enc_file_roundtrip_and_knn_fields(tmp_path):
    enc = np.random.default_rng(1).random((9, 3), dtype=np.float32)
    path = tmp_path / '5009.enc'
    schema.write_enc_file(path, enc)
    assert np.array_equal(schema.read_enc_file(path), enc)
    schema.write_enc_file(path, enc.tolist())                     # nested lists are legal .enc content too
    assert np.allclose(schema.read_enc_file(path), enc)
    import pickle
    with open(path, 'wb') as f:                                     # a pickle that would run code is refused
        pickle.dump(print, f)
    with pytest.raises(pickle.UnpicklingError):
        schema.read_enc_file(path)
    idx = np.arange(18, dtype=np.int32).reshape(9, 2) % 9
    doc = schema.encits_document('e', 1, enc, list('abcdefghi'), knn=(idx, idx.astype(np.float32)))
    assert doc['nn'][0] == [1, 2] and len(doc['nd']) == 9


def test_legacy_directory_import_keeps_the_rstrip_id_rule(tmp_path):
    """reference import_encodings.py:66-77: ids are `name.rstrip('.enc')` — trailing CHARACTERS of {., e, n, c} go,
    not the suffix — and the coordinates are then read from `<id>.enc`."""
    from cellcomm_amd.recorder import MemoryDatabase
    assert schema.legacy_iteration_id('5009.enc') == '5009'
    assert schema.legacy_iteration_id('100e.enc') == '100'          # not '100e'
    assert schema.legacy_iteration_id('49909.enc') == '49909'
    assert schema.legacy_iteration_id('dec.enc') == 'd'
    rng = np.random.default_rng(3)
    barcodes = ['AAAC-1', 'AAAG-1', 'AACT-1', 'AAGT-1']
    encs = {name: rng.random((4, 5), dtype=np.float32) for name in ('5009.enc', '100.enc', '100e.enc')}
    for name, enc in encs.items():
        schema.write_enc_file(tmp_path / name, enc)
    (tmp_path / 'barcodes.tsv').write_text(''.join(b + '\n' for b in barcodes))
    enc_dir = tmp_path / 'encodings'
    enc_dir.mkdir()
    for name in encs:
        (tmp_path / name).rename(enc_dir / name)
    in_order = lambda d: sorted(os.listdir(d))
    docs = schema.legacy_import_directory(str(enc_dir), str(tmp_path / 'barcodes.tsv'), listdir=in_order)
    assert [d['_id'] for d in docs] == ['100', '100', '5009']       # '100e.enc' collapsed onto '100' ...
    assert docs[0] == docs[1] == schema.legacy_encs_document('100', encs['100.enc'], barcodes)   # ... and read 100.enc
    assert docs[2]['xs'] == (encs['5009.enc'][:, 0] * np.float32(255)).tolist() and docs[2]['pids'] == [1, 2, 3, 4]
    assert docs[2]['ns'] == barcodes and set(docs[2]) == {'_id', 'pids', 'ns', 'xs', 'ys', 'zs'}
    db = MemoryDatabase()
    with pytest.raises(KeyError):                                    # ... so a real collection refuses the second '100'
        schema.legacy_import_directory(str(enc_dir), barcodes, collection=db['encs'], listdir=in_order)
    db = MemoryDatabase()
    one = schema.legacy_import_directory(str(enc_dir), barcodes, collection=db['encs'], listdir=lambda d: ['5009.enc'])
    assert db['encs'].docs == one == docs[2:]
    (enc_dir / '100.enc').unlink()                                   # now '100e.enc' points at a file that is not there
    with pytest.raises(FileNotFoundError):
        schema.legacy_import_directory(str(enc_dir), barcodes)
    with pytest.raises(ValueError) as e:                             # the length check of import_encodings.py:57-58
        schema.legacy_import_directory(str(enc_dir), barcodes[:3], listdir=lambda d: ['5009.enc'])
    assert str(e.value) == 'Coordinates-/barcode-data different length: 3 != 4'


# ---- BiGAN wrappers ----------------------------------------------------------------
def _mock_bigan():
    gen, enc, disc = MagicMock(), MagicMock(), MagicMock()
    b = bigan.BasicBiGan(encoding_size=15, gene_size=1000, generator_factory=MagicMock(return_value=gen),
                         encoder_factory=MagicMock(return_value=enc), discriminator_factory=MagicMock(return_value=disc))
    return b, gen, enc, disc


def test_create_components():
    b, gen, enc, disc = _mock_bigan()
    assert (b._generator, b._encoder, b._discriminator) == (gen, enc, disc)
    assert b.all_components == (gen, enc, disc) and b.encoding_size == 15


def test_factories_called_with_sizes():
    gf, ef, df = MagicMock(), MagicMock(), MagicMock()
    bigan.BasicBiGan(7, 33, gf, ef, df)
    for f in (gf, ef, df):
        f.assert_called_once_with(7, 33)


def test_encoding_prediction_is_pure_delegation(known):
    ka = known['encoding_prediction_delegation']
    b, _, enc, _ = _mock_bigan()
    enc.predict = MagicMock(return_value=ka['prediction'])
    assert b.encoding_prediction(ka['genes']) is enc.predict.return_value
    enc.predict.assert_called_with(ka['genes'])


def test_params_change_probe(capsys):
    class Layer:
        def __init__(self, ws): self.ws = ws
        def get_weights(self): return self.ws
    comps = [MagicMock(layers=[Layer([np.ones((2, 2)), np.zeros(2)]), Layer([np.ones(3)])]) for _ in range(3)]
    gan = bigan.BasicBiGan(4, 9, lambda e, g: comps[0], lambda e, g: comps[1], lambda e, g: comps[2])
    gan.print_params_changes('it 1')
    assert capsys.readouterr().out == 'it 1 G|E|D changed: 0 | 0 | 0,\n'
    comps[1].layers[0].ws[1][0] = 5.0
    gan.print_params_changes('it 2')
    assert capsys.readouterr().out == 'it 2 G|E|D changed: 0 | 1 | 0,\n'
    assert bigan.components_changed([np.ones(2)], [np.ones(2)]) is False
    assert bigan.components_changed([np.ones(2), np.zeros(1)], [np.ones(2), np.ones(1)]) is True


def test_random_uniform_vector_shape():
    b, *_ = _mock_bigan()
    hv = b.random_uniform_vector(7)
    assert hv.shape == (7, 15) and np.all(hv >= 0) and np.all(hv < 1)


def test_generate_cells_rounds_injected_generator():
    b, gen, _, _ = _mock_bigan()
    gen.predict = MagicMock(return_value=[[0.3, 12.59939265, 2.4894546, 0.01], [0.9, 4.7007282, 0, 2.07244989]])
    cells = b.generate_cells([[1, 0], [0, 1]], [[0.2, 0.], [0.99, 0.99]])
    assert np.array_equal(cells, [[0, 13, 2, 0], [1, 5, 0, 2]])
    gen.predict.assert_called_with(([[1, 0], [0, 1]], [[0.2, 0.], [0.99, 0.99]]))


def test_absent_components_fail_loudly():
    comp = bigan.AbsentComponent('cell_generator', 3, 5)
    assert comp.layers == []
    with pytest.raises(NotImplementedError):
        comp.predict(1)


def test_onehot_and_random_encoding_known_answers(known):
    enc_factory = MagicMock()
    cls = bigan.ClassifyCellBiGan(2, 4, encoder_factory=enc_factory)
    ka = known['trainings_encoding_prediction_onehot']
    cls.encoding_prediction = MagicMock(return_value=ka['prediction'])
    assert np.array_equal(cls.trainings_encoding_prediction([[5, 3, 1, 4]]), np.array(ka['onehot']))
    kr = known['random_encoding_vector']
    np.random.seed(kr['np_seed'])
    cls4 = bigan.ClassifyCellBiGan(kr['encoding_size'], 1, encoder_factory=MagicMock())
    assert np.array_equal(cls4.random_encoding_vector(kr['batch']), np.array(kr['expected']))


def test_layer_plan_matches_reference_graph(known):
    assert hidden_widths('cont', 30000) == [3000, 1500] and hidden_widths('cont', 5) == [0, 0]
    assert hidden_widths('cont', 11) == [1, 0] and hidden_widths('cls', 77) == [1000, 300]
    cont = _layer_plan('cont', 12, 100)
    cls = _layer_plan('cls', 12, 100)
    assert cont[-1][2] == known['encoder_shapes']['cont']['last_activation']
    assert cls[-1][2] == known['encoder_shapes']['cls']['last_activation']
    assert [l[1] for l in cont].count('Dense') == 5 and [l[1] for l in cls].count('Dense') == 4
    assert sorted(s for l in cont for s in l[3]) == list(range(14))
    assert sorted(s for l in cls for s in l[3]) == list(range(8))


# ---- sharding ------------------------------------------------------------------------
@pytest.mark.parametrize('n,world', [(100000, 8), (10, 4), (3, 8), (0, 2), (1000000, 8), (17, 1)])
def test_shard_bounds_partition(n, world):
    blocks = [pipeline.shard_bounds(n, world, r) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for (a, b), (c, d) in zip(blocks, blocks[1:]):
        assert b == c and a <= b and c <= d
    per = -(-n // world)
    assert all(hi - lo == per for lo, hi in blocks[:-1] if hi < n)


def test_native_parser_random_files_equal_pivot(tmp_path):
    # seeded sweep: ragged ids, duplicates, zeros, negative and float values, shuffled order
    rng = np.random.default_rng(77)
    for case in range(25):
        n_trip = int(rng.integers(1, 400))
        genes = rng.integers(-5, 60, n_trip) * int(rng.integers(1, 1000))
        cells = rng.integers(0, 30, n_trip) + int(rng.integers(0, 10 ** 9))
        if case % 3 == 0:
            vals = rng.integers(-3, 50, n_trip).astype(object)
        else:
            vals = np.round(rng.standard_normal(n_trip) * 10, int(rng.integers(0, 4))).astype(object)
        path = tmp_path / f'r{case}.mtx'
        sep = [' ', '\\t', '  '][case % 3]
        with open(path, 'w') as f:
            f.write('x\\ny\\nz\\n')
            for g, c, v in zip(genes, cells, vals):
                f.write(f'{g}{sep}{c}{sep}{v}\\n')
        want = lo.load_matrix(path)
        got = matrix.load_matrix(path)
        assert got.shape == want.shape, case
        assert np.array_equal(got.index, want.index.to_numpy()) and np.array_equal(got.columns, want.columns.to_numpy())
        assert np.array_equal(got.to_dense(np.float64), want.to_numpy().astype(np.float32).astype(np.float64)), case


def test_generator_plan_and_oracle_shapes():
    from cellcomm_amd import generator
    for arch, count in (('cont', 20), ('cls', 14)):
        plan = generator._layer_plan(arch, 13, 100)
        assert sorted(s for l in plan for s in l[3]) == list(range(count))
        assert plan[-1][2] == 'relu' and plan[-1][4] == 100                 # reference test/bigans_cc_test.py:24-25,98-99
        assert [l[0] for l in plan[:2]] == ['gen_encoding_in', 'gen_random_in']   # :19-23
        ws = eo_mod.generator_weights(arch, 13, 100, seed=1)
        assert [w.shape for w in ws] == [tuple(s) for s in eo_mod.generator_weight_shapes(arch, 13, 100)]
        out = eo_mod.generator_forward(arch, ws, np.zeros((3, 13)), np.ones((3, 13)))
        assert out.shape == (3, 100) and np.all(out >= 0)
