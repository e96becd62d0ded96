"""One parity test per BASELINE.json config, at the config's own shapes (oracle-checkable subsets
plus size-independent properties where the full oracle would take too long)."""
import numpy as np
import pytest

import cellcomm_amd as cc
from cellcomm_amd import synth
from cellcomm_amd.knn import knn_graph_device
from oracle import encoder_oracle as eo
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


def max_rel(got, truth):
    return float(np.max(np.abs(got.astype(np.float64) - truth) / np.abs(truth)))


def encode_range(enc, m, lo, hi):
    """Rows [lo, hi) of a matrix in storage order (a contiguous range: the sparse first layer's route)."""
    import torch
    out = torch.empty((max(hi - lo, 1), enc.encoding_size), dtype=torch.float32, device='cuda')
    enc.predict_to_device(m, out.data_ptr(), lo, hi - lo)
    return out.cpu().numpy()[:hi - lo]


def test_config2_encode_only_10k_x_20k_latent32(hip):
    # configs[1]: synthetic 10k cells x 20k genes, latent 32, encode only
    m, (indptr, indices, values) = synth.synth_csr_device(10000, 20000, seed=7)
    enc = cc.HipEncoder(32, 20000, 'cont', seed=4321)
    out = enc.predict(m)
    assert out.shape == (10000, 32) and np.all(out > 0) and np.all(out < 1)
    host = cc.CellMatrix(indptr.cpu().numpy(), indices.cpu().numpy(), values.cpu().numpy(), 20000)
    rows = np.random.default_rng(0).choice(10000, 64, replace=False)
    truth = eo.forward('cont', enc.get_weights(), host.take(rows).to_dense())
    assert max_rel(out[rows], truth) <= 1e-4
    assert np.array_equal(enc.predict(host.take(rows)), out[rows])      # row results are batch independent


@pytest.fixture(scope='module')
def c3(hip):
    """configs[2] at full size, shared by the tests below: the matrix on the device and on the host, the dense fp32
    encodings of all rows, 48 sampled rows with their float64 oracle truth."""
    n, g, e = 100000, 30000, 64
    m, (indptr, indices, values) = synth.synth_csr_device(n, g, seed=1234)
    enc = cc.HipEncoder(e, g, 'cont', seed=4321)
    out = enc.predict(m)
    host = cc.CellMatrix(indptr.cpu().numpy(), indices.cpu().numpy(), values.cpu().numpy(), g)
    rows = np.random.default_rng(3).choice(n, 48, replace=False)
    truth = eo.forward('cont', enc.get_weights(), host.take(rows).to_dense())
    yield dict(n=n, g=g, e=e, m=m, keep=(indptr, indices, values), enc=enc, out=out, host=host, rows=rows, truth=truth)


def test_config3_headline_size_properties(hip, oracle_built, c3):
    # configs[2], the bench workload at full size (100k cells x 30k genes, latent 64, k = 15): oracle on seeded
    # samples plus properties that do not need it — idempotence, independence of a row's encoding from the rows
    # around it, sortedness / self-exclusion / symmetry of the neighbour lists, f16x2 within tolerance of fp32.
    n, g, e, k = c3['n'], c3['g'], c3['e'], 15
    m, enc, out, host, rows, truth = c3['m'], c3['enc'], c3['out'], c3['host'], c3['rows'], c3['truth']
    rng = np.random.default_rng(3)
    assert out.shape == (n, e) and np.all(np.isfinite(out)) and np.all(out > 0) and np.all(out < 1)
    assert np.array_equal(enc.predict(m), out)                                    # idempotent, bit for bit
    assert max_rel(out[rows], truth) <= 1e-4
    block = host.take(np.arange(36990, 42010))                                    # straddles pass boundaries of the full run
    assert np.array_equal(enc.predict(block), out[36990:42010])
    # calls of at most two default passes take other plans (round 4): ONE pass of all rows whose partial last rounds run
    # as half-height tiles (10 000 rows: 2.47 + 1.23 rounds), or — when layer 1 would end in a nearly empty round (a
    # rank's 12 500 rows at 8 GPUs: 3.06) — the rows of the whole rounds on one stream and the remainder on the other.
    # Same MFMAs in the same order per output element: the same bits as inside the 100k-row call.
    for lo, hi in ((0, 12500), (87500, 100000), (40000, 50000), (3, 8195), (70000, 70064)):
        assert np.array_equal(encode_range(enc, m, lo, hi), out[lo:hi]), (lo, hi)
    enc2 = cc.HipEncoder(e, g, 'cont', seed=4321, dtype='f16x2')
    out2 = enc2.predict(m)
    assert enc2.fp32_fallback_passes() == 0
    assert max_rel(out2[rows], truth) <= 1e-4 and max_rel(out2, out.astype(np.float64)) <= 2e-4
    idx, dist = cc.knn_graph(out, k)
    assert np.array_equal(cc.knn_graph(out, k)[0], idx)
    assert np.all(idx != np.arange(n)[:, None]) and np.all(idx >= 0) and np.all(idx < n)
    assert np.all(np.diff(dist, axis=1) >= 0)
    for q in rng.choice(n, 48, replace=False):
        wi, wd = ko.knn(out, k, q_begin=int(q), n_q=1)
        assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0])
    # symmetry: if j is i's nearest neighbour and i is listed by j, both report the same distance
    j = idx[:, 0]
    back = (idx[j] == np.arange(n)[:, None])
    has = back.any(axis=1)
    assert has.mean() > 0.3
    assert np.array_equal(dist[j][back][:], dist[has, 0])


def test_config3_sparse_first_layer_at_the_size_the_bench_times_it(hip, c3):
    # configs[2] through the sparse first layer (bench `alt_sparse`: 100k rows = 391 row blocks x 12 panels, stream
    # offsets near 1.5e8, two 65536-row passes): sampled rows against the float64 oracle, ALL rows against the dense
    # encoder, idempotence, and row ranges that straddle 256-row blocks and the pass seam equal to the full call.
    n, g, e = c3['n'], c3['g'], c3['e']
    m, out, rows, truth = c3['m'], c3['out'], c3['rows'], c3['truth']
    sp = cc.HipEncoder(e, g, 'cont', seed=4321, sparse=True)
    got = sp.predict(m)
    assert got.shape == (n, e) and np.all(np.isfinite(got))
    assert max_rel(got[rows], truth) <= 1e-4
    assert max_rel(got, out.astype(np.float64)) <= 2e-5                 # same fp32 sums in another order
    assert np.array_equal(sp.predict(m), got)                           # idempotent, bit for bit
    for lo, hi in ((256 * 100 - 7, 256 * 103 + 5), (65536 - 300, 65536 + 300), (99000, n), (0, 1), (n - 1, n)):
        assert np.array_equal(encode_range(sp, m, lo, hi), got[lo:hi]), (lo, hi)
    few = cc.HipEncoder(e, g, 'cont', seed=4321, sparse=True, batch_rows=1024)   # 8192-row sparse passes
    assert np.array_equal(encode_range(few, m, 30000, 50000), got[30000:50000])


@pytest.mark.parametrize('dtype', ['f16', 'f16x2'])
def test_config3_fp16_whole_rounds_equal_the_half_height_tiles(hip, c3, dtype):
    """The fp16 kernels have two K loops: whole rounds of 256 x 256 tiles run the ping-pong of the two wave groups
    (counted waits on LDS-DMA, csrc/gemm16.hip), a last round that is at most half full — and every call of a few
    thousand rows — the 128-row tiles with the older loop.  Same MFMAs in the same order per accumulator: ALL rows of a
    65536-row call (two passes of six whole rounds) must equal, bit for bit, the same rows encoded 2048 at a time
    (96 layer-1 tiles: half-height only), and a second run of the long call.  A misplaced wait in the ping-pong shows
    up here as a rare wrong tile."""
    m, e, g = c3['m'], c3['e'], c3['g']
    enc = cc.HipEncoder(e, g, 'cont', seed=4321, dtype=dtype)
    n = 65536
    whole = encode_range(enc, m, 0, n)
    assert np.array_equal(encode_range(enc, m, 0, n), whole)
    for lo in range(0, n, 2048):
        part = encode_range(enc, m, lo, lo + 2048)
        assert np.array_equal(part, whole[lo:lo + 2048]), f'{dtype}: rows {lo}..{lo + 2048} differ between the two K loops'
    sel = c3['rows'] < n                                            # the sampled rows that fall into the range
    rows = c3['rows'][sel]
    truth = c3['truth'][sel] if dtype == 'f16x2' else eo.forward_f16('cont', enc.get_weights(), c3['host'].take(rows).to_dense())
    assert sel.sum() >= 16 and max_rel(whole[rows], truth) <= 1e-4
    assert enc.fp32_fallback_passes() == 0


def test_config5_fp16_encoder_at_one_ranks_shape(hip):
    # configs[4]: one rank's encode, 62 500 cells x 33 000 genes with fp16 operands (bench --dtype f16 / `alt_f16`):
    # ONE pass at this width's default (65536 rows: 13 + 7 column tiles fill whole rounds; the partial last round of the
    # 62 500 rows runs as half-height tiles).  Sampled rows against the float64 evaluation of the once-rounded operands;
    # the same call in 32768-row passes (the usual fp16 pass: a seam inside the range) returns the same bits, and the
    # rows around that seam equal the same rows encoded alone; idempotence.
    n, g, e = 62500, 33000, 64
    m, (indptr, indices, values) = synth.synth_csr_device(n, g, seed=55)
    enc = cc.HipEncoder(e, g, 'cont', seed=4321, dtype='f16')
    out = enc.predict(m)
    assert enc.fp32_fallback_passes() == 0
    assert out.shape == (n, e) and np.all(np.isfinite(out)) and np.all(out > 0) and np.all(out < 1)
    assert np.array_equal(enc.predict(m), out)
    host = cc.CellMatrix(indptr.cpu().numpy(), indices.cpu().numpy(), values.cpu().numpy(), g)
    rows = np.concatenate([np.random.default_rng(8).choice(n, 44, replace=False), [0, 32767, 32768, n - 1]])
    assert max_rel(out[rows], eo.forward_f16('cont', enc.get_weights(), host.take(rows).to_dense())) <= 1e-4
    assert enc.pass_rows == 65536
    two = cc.HipEncoder(e, g, 'cont', seed=4321, dtype='f16', batch_rows=32768)
    assert np.array_equal(two.predict(m), out)
    seam = np.arange(32760, 32777)
    for each in (enc, two):
        assert np.array_equal(each.predict(host.take(seam)), out[seam])
        assert np.array_equal(encode_range(each, m, 32760, 32777), out[32760:32777])


def test_config4_shard_of_1m_candidates(hip, oracle_built):
    # configs[3]: 1M cells sharded over 8 GPUs — one rank's kNN: its 125k queries against all 1M
    # encodings, global indices; 48 seeded queries checked bit-exact against the oracle.
    import torch
    gen = torch.Generator(device='cuda')
    gen.manual_seed(5)
    x = torch.sigmoid(torch.randn((1000000, 64), device='cuda', generator=gen))
    torch.cuda.synchronize()
    q_begin, n_q = 875000, 125000
    idx, dist = knn_graph_device(x.data_ptr(), 1000000, 64, 15, q_begin=q_begin, n_q=n_q)
    host = x.cpu().numpy()
    for ql in np.random.default_rng(1).choice(n_q, 48, replace=False):
        wi, wd = ko.knn(host, 15, q_begin=q_begin + int(ql), n_q=1)
        assert np.array_equal(idx[ql], wi[0]) and np.array_equal(dist[ql], wd[0])
    assert np.all(idx != (q_begin + np.arange(n_q))[:, None]) and np.all(np.diff(dist, axis=1) >= 0)


def test_config4_one_ranks_encode_125k_rows(hip):
    # configs[3]: one rank's encode (125k of the 1M cells, G = 30000, latent 64).  Full-size truth is not affordable on
    # the host, so: 48 sampled rows against the float64 oracle, plus properties — idempotence, independence of a row
    # from its pass (8192-row passes: 15 full + 1 of 2120 rows) and from the call it sits in.
    n, g, e = 125000, 30000, 64
    m, (indptr, indices, values) = synth.synth_csr_device(n, g, seed=77)
    enc = cc.HipEncoder(e, g, 'cont', seed=4321)
    out = enc.predict(m)
    assert out.shape == (n, e) and np.all(np.isfinite(out)) and np.all(out > 0) and np.all(out < 1)
    assert np.array_equal(enc.predict(m), out)
    host = cc.CellMatrix(indptr.cpu().numpy(), indices.cpu().numpy(), values.cpu().numpy(), g)
    rows = np.random.default_rng(4).choice(n, 48, replace=False)
    assert max_rel(out[rows], eo.forward('cont', enc.get_weights(), host.take(rows).to_dense())) <= 1e-4
    tail = host.take(np.arange(122000, n))                       # crosses the last full pass into the short one
    assert np.array_equal(enc.predict(tail), out[122000:])
    assert np.array_equal(enc.predict(host.take(rows)), out[rows])


def test_config5_one_ranks_knn_62500_queries_of_500k_k50(hip, oracle_built):
    # configs[4]: 500k cells over 8 GPUs, k = 50 — one rank's REAL search shape: its 62 500 queries against all
    # 500 000 candidates (only 62.5k x 62.5k had been run).  48 sampled queries bit-exact against
    # oracle/knn_oracle.c, plus properties over all 62 500 lists.
    import torch
    n, d, k = 500000, 64, 50
    gen = torch.Generator(device='cuda')
    gen.manual_seed(21)
    x = torch.sigmoid(torch.randn((n, d), device='cuda', generator=gen))
    x[250000:250040] = x[123]                                     # a run of exact duplicates inside the query block
    torch.cuda.synchronize()
    q_begin, n_q = 4 * 62500, 62500                               # rank 4 of 8
    idx, dist = knn_graph_device(x.data_ptr(), n, d, k, q_begin=q_begin, n_q=n_q)
    idx, dist = idx.copy(), dist.copy()
    host = x.cpu().numpy()
    picks = np.concatenate([np.random.default_rng(6).choice(n_q, 44, replace=False), [0, 1, 39, 40]])
    for ql in picks:
        wi, wd = ko.knn(host, k, q_begin=q_begin + int(ql), n_q=1)
        assert np.array_equal(idx[ql], wi[0]) and np.array_equal(dist[ql], wd[0])
    qid = q_begin + np.arange(n_q)
    assert np.all(idx != qid[:, None]) and np.all(idx >= 0) and np.all(idx < n)
    assert np.all(np.diff(dist, axis=1) >= 0)
    # exact ties are ordered by index (float32 outputs may tie where the float64 distances that decide the order
    # do not, so the property is checked where the tie is certain: distance 0)
    zero_ties = (dist[:, 1:] == 0) & (dist[:, :-1] == 0)
    assert np.all(np.diff(idx, axis=1)[zero_ties] > 0)
    assert np.all(dist[:40, :39] == 0) and np.all(dist[:40, 39] == 0)   # the 40 duplicates + row 123 see each other at 0
    again, _ = knn_graph_device(x.data_ptr(), n, d, k, q_begin=q_begin, n_q=n_q)
    assert np.array_equal(again, idx)


def test_config5_fp16_encoder_33k_genes_and_k50(hip, oracle_built):
    # configs[4]: cont encoder at G = 33000 with fp16 MFMA operands, k = 50 kNN
    g, e = 33000, 64
    m = synth.synth_csr(40, g, seed=12)
    enc = cc.HipEncoder(e, g, 'cont', seed=4321, dtype='f16')
    ws = enc.get_weights()
    out = enc.predict(m)
    assert max_rel(out, eo.forward_f16('cont', ws, m.to_dense())) <= 1e-4
    import torch
    gen = torch.Generator(device='cuda')
    gen.manual_seed(9)
    x = torch.rand((62500, e), device='cuda', generator=gen)             # one of 8 shards of 500k
    torch.cuda.synchronize()
    idx, dist = knn_graph_device(x.data_ptr(), 62500, e, 50)
    host = x.cpu().numpy()
    for q in np.random.default_rng(2).choice(62500, 64, replace=False):
        wi, wd = ko.knn(host, 50, q_begin=int(q), n_q=1)
        assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0])
