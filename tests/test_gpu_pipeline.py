"""End to end on one GPU through the reference's call surface."""
import numpy as np
import pytest

import cellcomm_amd as cc
from cellcomm_amd import pipeline, schema, synth
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


def test_trainer_surface_on_example_matrix(hip, golden_dir, known):
    data = cc.load_matrix(f'{golden_dir}/example_matrix.mtx')
    ka = known['trainer_setup']
    trainer = cc.CellTraining(data, 3, ka['encoding_size'], batches_per_iteration=4)
    net = trainer.network
    assert net.encoding_size == ka['encoding_size'] and net._encoder.input_shape == (None, ka['gene_count'])
    enc = net.encoding_prediction(trainer.data)                 # the call at reference db_recorder.py:85
    assert enc.shape == (5, 8) and enc.dtype == np.float32
    truth = eo.forward('cont', net._encoder.get_weights(), data.to_dense())
    assert np.max(np.abs(enc - truth) / np.abs(truth)) <= 1e-4
    batch = trainer.sample_cell_data(0)
    assert np.array_equal(net.trainings_encoding_prediction(batch), enc[[2, 0, 1]])
    net.summary()
    assert net.params_changed() == [False, False, False]
    with pytest.raises(NotImplementedError):
        trainer.run(1)


def test_encits_document_from_gpu_encodings(hip, golden_dir):
    data = cc.load_matrix(f'{golden_dir}/example_matrix.mtx')
    net = cc.ContinuousCellBiGan(3, data.shape[1])
    enc = net.encoding_prediction(data)
    barcodes = schema.load_barcodes(f'{golden_dir}/example_barcodes.tsv')
    doc = schema.encits_document('run', 5009, enc, barcodes, knn=cc.knn_graph(enc, 2))
    # G=5 -> zero-width hidden layers -> all cells share one encoding -> one duplicate group of all ids
    assert doc['cids'] == [1, 2, 3, 4, 5] and doc['ds'] == [[1, 2, 3, 4, 5]]
    assert doc['nn'][0] == [2, 3] and doc['nn'][4] == [1, 2]


@pytest.mark.parametrize('arch', ['cont', 'cls'])
def test_encode_and_knn_single_gpu(hip, oracle_built, arch):
    n, g, e, k = 1500, 2000, 32, 15
    m = synth.synth_csr(n, g, seed=21)
    enc = cc.HipEncoder(e, g, arch, seed=4321)
    res = pipeline.encode_and_knn(enc, m, k)
    got = res['encodings'].cpu().numpy()
    truth = eo.forward(arch, enc.get_weights(), m.to_dense())
    assert np.max(np.abs(got - truth) / np.abs(truth)) <= 1e-4
    want_i, want_d = ko.knn(got, k)                             # bit-exact on identical inputs
    assert np.array_equal(res['idx'], want_i) and np.array_equal(res['dist'], want_d)
    assert res['q_begin'] == 0 and res['n_total'] == n


def test_host_copy_of_encodings_under_the_knn(hip, oracle_built):
    import torch
    n, g, e, k = 3000, 2000, 32, 15
    m = synth.synth_csr(n, g, density=0.05, seed=21)
    enc = cc.HipEncoder(e, g, 'cont', seed=2)
    pinned = torch.empty((n, e), dtype=torch.float32, pin_memory=True)
    res = pipeline.encode_and_knn(enc, m, k, host_encodings=pinned)
    assert np.array_equal(pinned.numpy(), res['encodings'].cpu().numpy())
    assert np.array_equal(pinned.numpy(), enc.predict(m))


def test_device_resident_csr(hip, oracle_built):
    m_dev, (indptr, indices, values) = synth.synth_csr_device(700, 3000, seed=5)
    host = cc.CellMatrix(indptr.cpu().numpy(), indices.cpu().numpy(), values.cpu().numpy(), 3000)
    assert np.all(host.values >= 1) and np.all(host.values == np.round(host.values))
    assert abs(host.nnz / (700 * 3000) - 0.05) < 0.005
    enc = cc.HipEncoder(8, 3000, 'cont', seed=2)
    assert np.array_equal(enc.predict(m_dev), enc.predict(host))


def test_duplicate_groups_on_gpu_equal_reference(hip):
    # the `ds` field: GPU grouping + reference group order == oracle restatement of find_duplicate_ids
    from oracle import schema_oracle as so
    rng = np.random.default_rng(13)
    enc = rng.integers(0, 5, (3000, 3)).astype(np.float32) / 4            # saturated lattice: many duplicates
    enc[7] = [0.0, -0.0, 0.5]
    enc[9] = [-0.0, 0.0, 0.5]                                              # -0.0 == +0.0
    coords = np.multiply(enc, 255)
    got = schema.find_duplicate_ids_gpu(coords)
    assert got == so.find_duplicate_ids(coords)
    assert got == schema.find_duplicate_ids(coords)
    assert any(8 in g and 10 in g for g in got)
    # NaN equals nothing (and CPython >= 3.10 hashes a NaN by object identity, so with NaN rows the
    # reference's own group ORDER changes from run to run): contents only
    enc[11] = [np.nan, 0.25, 0.25]
    enc[12] = [np.nan, 0.25, 0.25]
    coords = np.multiply(enc, 255)
    got = schema.find_duplicate_ids_gpu(coords)
    assert sorted(got) == sorted(so.find_duplicate_ids(coords))
    assert not any(12 in g or 13 in g for g in got)
    big = np.multiply(rng.integers(0, 60, (100000, 3)).astype(np.float32) / 59, 255)
    assert schema.find_duplicate_ids_gpu(big) == schema.find_duplicate_ids(big)
    assert schema.find_duplicate_ids_gpu(np.zeros((0, 3), dtype=np.float32)) == []
    uniq = np.multiply(rng.random((5000, 3), dtype=np.float32), 255)
    assert schema.find_duplicate_ids_gpu(uniq) == []


def test_integration_md_stub_runs_as_written(hip):
    # the ctypes stub printed in INTEGRATION.md, executed verbatim against the built library
    import re
    from conftest import ROOT
    from cellcomm_amd import _lib
    text = open(f'{ROOT}/INTEGRATION.md').read()
    code = re.search(r'```python\n(import ctypes, numpy as np.*?)```', text, flags=re.S).group(1)
    code = code.replace('/path/to/libcellcomm_hip.so', _lib.lib_path())
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    g, e = 300, 6
    ws = eo.glorot_weights('cont', e, g, seed=3, randomize_bn=True, bias_scale=0.1)

    class KerasLike:                                   # what the reference's _build_encoder returns, as far as the stub cares
        def get_weights(self):
            return ws

    stub = ns['hip_cont_encoder_factory'](lambda es, gs: KerasLike())(e, g)
    x = np.random.default_rng(0).integers(0, 5, (17, g)).astype(np.float32)
    out = stub.predict(x.tolist())
    truth = eo.forward('cont', ws, x)
    assert out.shape == (17, e) and np.max(np.abs(out - truth) / np.abs(truth)) <= 1e-4
    assert len(stub.layers) == 5 and len(stub.layers[0].get_weights()) == 2
    net = cc.BasicBiGan(e, g, lambda a, b: cc.bigan.AbsentComponent('g', a, b), lambda a, b: stub,
                        lambda a, b: cc.bigan.AbsentComponent('d', a, b))
    assert np.array_equal(net.encoding_prediction(x), out)
