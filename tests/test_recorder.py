"""The run recorder (reference src/intercepts/db_recorder.py) against the known answers the reference's own
test/db_recorder_test.py holds — restated without a Mongo server (MemoryDatabase implements the five
collection calls the reference makes; a pymongo database can be passed instead)."""
import json
import os
from datetime import datetime
from types import SimpleNamespace

import numpy as np
import pytest

from cellcomm_amd import recorder, schema
from cellcomm_amd.recorder import DbRecorder, MemoryDatabase


@pytest.fixture
def sources(golden_dir):
    return {'matrix': os.path.join(golden_dir, 'example_matrix.mtx'),
            'barcodes': os.path.join(golden_dir, 'example_barcodes.tsv'),
            'genes': os.path.join(golden_dir, 'example_genes.tsv')}


class FixedNetwork:
    def __init__(self, encodings):
        self.encodings, self.calls = encodings, []

    def encoding_prediction(self, data):
        self.calls.append(data)
        return self.encodings


def trainer_returning(encodings):
    return SimpleNamespace(network=FixedNetwork(np.array(encodings)), data='the-data')


def test_check_files(known, sources):
    bad = dict(sources, barcodes=os.path.join(os.path.dirname(sources['barcodes']), 'does.not.exist'))
    with pytest.raises(AssertionError) as err:
        DbRecorder('fail-run-id', bad)
    assert str(err.value) == known['recorder']['file_not_found'].format(path=bad['barcodes'])


def test_stores_encoding_run(known, sources):
    want = known['encs_run_document']
    db = MemoryDatabase()
    DbRecorder(want['run_id'], sources, db).store_encoding_run()
    doc = db[recorder.ENCODINGS_COLLECTION].find_one({'_id': want['run_id']})
    assert doc['_id'] == want['run_id']
    assert (datetime.now() - doc['date']).total_seconds() < 1
    assert doc['defit'] == want['defit'] and doc['showits'] == want['showits']
    assert doc['srcs'] == want['srcs']
    assert set(doc) == {'_id', 'date', 'defit', 'showits', 'srcs'}     # what cellan's /encoding/:id returns


def test_encs_run_document_builder(known, sources):
    want = known['encs_run_document']
    now = datetime(2020, 5, 17, 12, 0, 0)
    doc = schema.encs_run_document(want['run_id'], sources, now)
    assert doc == {'_id': want['run_id'], 'date': now, 'defit': 0, 'showits': [], 'srcs': want['srcs']}


def test_encoding_run_exists(known, sources):
    rec = DbRecorder('test-run', sources)
    rec.store_encoding_run()
    with pytest.raises(AssertionError) as err:
        rec.store_encoding_run()
    assert str(err.value) == known['recorder']['run_exists']


def test_stores_barcodes_in_cells_and_genes(known, sources):
    k = known['recorder']
    db = MemoryDatabase()
    rec = DbRecorder('test-run', sources, db)
    rec.setup()
    sid = 'example_barcodes.tsv'
    cells = db[recorder.CELLS_COLLECTION].find({'sid': sid})
    assert len(cells) == 5
    for cell, want in ((cells[0], k['first_cell']), (cells[4], k['last_cell'])):
        assert cell['sid'] == sid and cell['cid'] == want['cid'] and cell['n'] == want['n']
        for got_gene, want_gene in zip(cell['g'], want['g']):
            assert got_gene == want_gene
    genes = db[recorder.GENES_COLLECTION].find({'sid': sid})
    assert len(genes) == 5
    assert genes[0] == dict(sid=sid, **k['first_gene'])
    assert genes[4] == dict(sid=sid, **k['last_gene'])
    assert rec.barcodes == known['encits_document']['ns'] and rec.cell_ids == [1, 2, 3, 4, 5]


def test_use_existing_barcodes_in_cells(known, sources):
    want = known['recorder']['existing_cells']
    db = MemoryDatabase()
    db[recorder.CELLS_COLLECTION].insert_many([{'sid': 'example_barcodes.tsv', 'n': n} for n in want['names']])
    rec = DbRecorder('test-run', sources, db)
    rec.setup()
    assert rec.barcodes == want['names'] and rec.cell_ids == want['cell_ids']
    assert db[recorder.GENES_COLLECTION].docs == []          # nothing imported when the source is already there


def test_order_of_calls_is_enforced(known, sources):
    rec = DbRecorder('test-run', sources)
    with pytest.raises(AssertionError) as err:
        rec.load_barcodes()
    assert rec.barcodes is None and str(err.value) == known['recorder']['no_encoding']
    with pytest.raises(AssertionError) as err:
        rec.create_interceptor(None)
    assert str(err.value) == known['recorder']['no_barcodes']


def test_intercept_stores_iteration_and_updates_run(known, sources):
    want, k = known['encits_document'], known['recorder']
    db = MemoryDatabase()
    rec = DbRecorder('test-run', sources, db)
    rec.setup()
    trainer = trainer_returning(want['encodings'])
    intercept = rec.create_interceptor(trainer)
    for it in k['iterations']:
        intercept(it, {'what': 'ever'})
    its = db[recorder.ITERATIONS_COLLECTION].find({'eid': 'test-run'}, {'_id': 0})
    assert len(its) == 2
    assert its[0] == {'eid': 'test-run', 'it': k['iterations'][0], 'cids': want['cids'], 'ns': want['ns'],
                      'xs': want['xs'], 'ys': want['ys'], 'zs': want['zs'], 'ds': want['ds']}
    assert trainer.network.calls == ['the-data', 'the-data']          # the full data set, every iteration
    run = db[recorder.ENCODINGS_COLLECTION].find_one({'_id': 'test-run'})
    assert run['defit'] == k['defit_after'] and run['showits'] == k['showits_after']


def test_intercept_assertions(known, sources):
    rec = DbRecorder('test-run', sources)
    rec.setup()
    with pytest.raises(AssertionError) as err:
        rec.create_interceptor(trainer_returning([[], [], [], []]))(2, None)
    assert str(err.value) == known['messages']['length']
    with pytest.raises(AssertionError) as err:
        rec.create_interceptor(trainer_returning([[1, 2]] * 5))(3, None)
    assert str(err.value) == known['messages']['xyz']


def test_duplicate_iteration(known, sources):
    rec = DbRecorder('test-run', sources)
    rec.setup()
    intercept = rec.create_interceptor(trainer_returning(known['encits_document']['encodings']))
    intercept(2009, None)
    with pytest.raises(AssertionError) as err:
        intercept(2009, None)
    assert str(err.value) == known['recorder']['duplicate_iteration']


def test_wide_encodings_export_three_dims(known, sources):
    want = known['encits_document']
    rng = np.random.RandomState(3)
    wide = rng.uniform(size=(5, 64)).astype(np.float32)
    wide[:, [0, 1, 2]] = np.array(want['encodings'], dtype=np.float32)
    doc = schema.encits_from_wide('run', 7, wide, want['ns'])
    assert doc['xs'] == want['xs'] and doc['zs'] == want['zs'] and doc['ds'] == want['ds']
    assert np.allclose(doc['ys'], want['ys'], rtol=1e-6)
    other = schema.encits_from_wide('run', 7, wide, want['ns'], dims=(5, 9, 63))
    assert np.array_equal(np.float32(other['zs']), wide[:, 63] * np.float32(255))
    with pytest.raises(AssertionError):
        schema.encits_from_wide('run', 7, wide, want['ns'], dims=(0, 1, 64))
    # through the recorder: the reference's width assertion stays the default
    rec = DbRecorder('wide-run', sources)
    rec.setup()
    with pytest.raises(AssertionError) as err:
        rec.create_interceptor(trainer_returning(wide))(1, None)
    assert str(err.value) == 'encodings vector length = 64, not in x, y, z format'
    rec.create_interceptor(trainer_returning(wide), dims=(0, 1, 2))(2, None)
    stored = rec.mongo_db[recorder.ITERATIONS_COLLECTION].find_one({'eid': 'wide-run', 'it': 2})
    assert stored['xs'] == want['xs'] and stored['ds'] == want['ds']


def test_memory_database_dump(tmp_path, sources, known):
    db = MemoryDatabase()
    rec = DbRecorder('test-run', sources, db)
    rec.setup()
    rec.create_interceptor(trainer_returning(known['encits_document']['encodings']))(5009, None)
    db.dump(tmp_path)
    run = [json.loads(ln) for ln in open(tmp_path / 'encs.json')]
    assert run[0]['_id'] == 'test-run' and run[0]['defit'] == 5009 and run[0]['showits'] == [5009]
    its = [json.loads(ln) for ln in open(tmp_path / 'encits.json')]
    assert its[0]['ds'] == known['encits_document']['ds']
    assert len(open(tmp_path / 'cells.json').readlines()) == 5
