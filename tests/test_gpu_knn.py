"""kNN graph on the GPU: neighbour indices bit-exact against oracle/knn_oracle.c."""
import numpy as np
import pytest

import cellcomm_amd as cc
from conftest import load_npz
from cellcomm_amd import knn as hipknn
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


def check(x, k, include_self=False):
    idx, dist = cc.knn_graph(x, k, include_self)
    want_i, want_d = ko.knn(x, k, include_self)
    assert idx.dtype == np.int32 and dist.dtype == np.float32
    assert np.array_equal(idx, want_i)
    assert np.array_equal(dist, want_d)
    return idx


@pytest.mark.parametrize('name', ['knn_N257_E3.npz', 'knn_N2048_E32.npz', 'knn_N2048_E64.npz'])
@pytest.mark.parametrize('k', [15, 50])
def test_golden_fixtures(hip, oracle_built, name, k):
    z = load_npz(name)
    idx, dist = cc.knn_graph(z['x'], k)
    assert np.array_equal(idx, z[f'idx{k}'])
    assert np.array_equal(dist, z[f'dist{k}'])


@pytest.mark.parametrize('n,d,k', [(1000, 1, 5), (999, 2, 15), (5000, 3, 15), (3001, 8, 1), (2500, 17, 33),
                                   (1200, 40, 64), (700, 100, 128), (600, 132, 15), (33, 64, 15), (2, 5, 1)])
def test_shapes_and_dims(hip, oracle_built, n, d, k):
    x = np.random.default_rng(n + d).random((n, d), dtype=np.float32)
    check(x, k)


@pytest.mark.parametrize('n,d,k', [(300, 64, 15), (512, 8, 50), (513, 8, 50), (1000, 64, 15), (2048, 3, 33), (6000, 64, 15),
                                   (1200, 40, 64), (1537, 16, 128), (5000, 64, 128)])
def test_small_inputs_both_routes(hip, oracle_built, n, d, k):
    # small inputs take the collect route too (without the sample when they fit the collect buffer whole);
    # the full select (stage 1a) must give the same bits
    rng = np.random.default_rng(n * 7 + d + k)
    x = rng.random((n, d), dtype=np.float32)
    x[rng.choice(n, n // 10, replace=False)] = x[rng.choice(n, n // 10)]
    idx = check(x, k)
    hipknn.force_full_select(True)
    try:
        idx_a = check(x, k)
    finally:
        hipknn.force_full_select(False)
    assert np.array_equal(idx, idx_a)


def test_include_self_and_signed_data(hip, oracle_built):
    x = np.random.default_rng(3).standard_normal((1500, 24)).astype(np.float32) * 3
    idx = check(x, 15, include_self=True)
    assert np.array_equal(idx[:, 0], np.arange(1500))
    check(x, 15, include_self=False)


def test_heavy_duplicates_need_the_fallback(hip, oracle_built):
    # 3000 cells on 20 distinct encodings: every query has hundreds of exact ties at distance 0,
    # far more than the K' survivors of stage 1 -> the certificate must fail and stage 3 decide.
    rng = np.random.default_rng(8)
    sites = rng.random((20, 16), dtype=np.float32)
    x = sites[rng.integers(0, 20, 3000)]
    check(x, 15)
    assert hipknn.last_fallback_count() > 0
    x3 = np.round(rng.random((4000, 3)) * 4).astype(np.float32) / 4      # saturated 3-d lattice (ds-style duplicates)
    check(x3, 50)


@pytest.mark.parametrize('d,k,csize,spread', [(64, 15, 500, 0.01), (64, 15, 40, 0.3), (30, 33, 100, 0.02), (7, 15, 300, 1e-3),
                                               (132, 5, 64, 0.05)])
def test_tight_sorted_clusters_take_the_float64_prefilter(hip, oracle_built, d, k, csize, spread):
    # all neighbours nearly equidistant and far from the centroid: the fp32/fp16 certificate fails for most
    # queries; stage 3a (float64 matrix-core prefilter) must leave only short candidate lists, and the result
    # must equal the oracle's and the one obtained with every uncertified query scanning all candidates
    n = 6000
    rng = np.random.default_rng(d * 1000 + k)
    nc = n // csize
    centres = rng.random((nc, d), dtype=np.float32) * 10
    lab = np.sort(rng.integers(0, nc, n))
    x = (centres[lab] + rng.standard_normal((n, d)).astype(np.float32) * np.float32(spread)).astype(np.float32)
    x[rng.choice(n, 60, replace=False)] = x[rng.choice(n, 60)]                 # some exact duplicates on top
    idx = check(x, k)
    n_fb, n_full = hipknn.last_fallback_count(), hipknn.last_full_scan_count()
    hipknn.use_fp64_prefilter(False)
    try:
        idx_b = check(x, k)
        assert hipknn.last_full_scan_count() == hipknn.last_fallback_count()
    finally:
        hipknn.use_fp64_prefilter(True)
    assert np.array_equal(idx, idx_b)
    if d == 64 and csize == 500:
        assert n_fb > n // 2                      # the case this stage exists for
    assert n_full <= n_fb // 10 + 5               # and nearly all of them are settled from the short lists


def test_forced_fallback_agrees(hip, oracle_built):
    x = np.random.default_rng(4).random((1800, 32), dtype=np.float32)
    x[100:140] = x[5]
    hipknn.force_fallback(True)
    try:
        check(x, 15)
        assert hipknn.last_fallback_count() == 1800
        check(x, 50, include_self=True)
    finally:
        hipknn.force_fallback(False)
    check(x, 15)
    assert hipknn.last_fallback_count() < 1800


@pytest.mark.parametrize('n_q,k', [(1, 15), (3, 50), (40, 128), (300, 15)])
def test_few_uncertified_queries_scan_in_parallel(hip, oracle_built, n_q, k):
    """A query without a usable prefilter list scans every candidate in float64; a handful of such queries are split
    over candidate ranges (one workgroup walking 500k candidates alone cost 6.4 ms) and merged.  Forced here with the
    fallback hook and the prefilter off on a small query range; duplicates sit across range boundaries."""
    import torch
    rng = np.random.default_rng(n_q + k)
    n, d = 60000, 24
    x = rng.random((n, d), dtype=np.float32)
    x[4090:4100] = x[20000]                                  # exact ties on both sides of a 4096-candidate boundary
    x[59990:] = x[20001]
    q0 = 19990
    t = torch.from_numpy(x).cuda()
    hipknn.force_fallback(True)
    hipknn.use_fp64_prefilter(False)
    try:
        idx, dist = hipknn.knn_graph_device(t.data_ptr(), n, d, k, q_begin=q0, n_q=n_q)
        assert hipknn.last_fallback_count() == n_q and hipknn.last_full_scan_count() == n_q
    finally:
        hipknn.force_fallback(False)
        hipknn.use_fp64_prefilter(True)
    want_i, want_d = ko.knn(x, k, q_begin=q0, n_q=n_q)
    assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d)


@pytest.mark.parametrize('kind', ['uniform', 'far_from_origin', 'tight_clusters', 'dup_heavy', 'tiny_scale', 'mixed_scale'])
def test_coarse_and_fp32_list_collect_agree(hip, oracle_built, kind):
    """The collect pass keeps a pair when ONE fp16 product puts it within T + its error bound and the direct fp32
    re-evaluation confirms it (default), or evaluates every pair on the fp32 matrix cores (option off: the one
    cross-check route; rounds 1-2 also had a three-product fp16 form, retired in round 3): the neighbour lists must be
    the oracle's either way, on data that stresses the coarse bound (large norms against small distances,
    denormal-range coordinates, exact duplicates)."""
    rng = np.random.default_rng(len(kind))
    n, d, k = 9000, 64, 15
    if kind == 'uniform':
        x = rng.random((n, d), dtype=np.float32)
    elif kind == 'far_from_origin':
        x = (rng.random((n, d)) * 1e-2 + 50.0).astype(np.float32)
        x[:4500, :8] += 30.0                                  # two far-apart groups: large centred norms, small distances
    elif kind == 'tight_clusters':
        c = rng.random((30, d), dtype=np.float32) * 10
        x = (c[np.sort(rng.integers(0, 30, n))] + rng.standard_normal((n, d)).astype(np.float32) * 0.01).astype(np.float32)
    elif kind == 'dup_heavy':
        sites = rng.random((200, d), dtype=np.float32)
        x = sites[rng.integers(0, 200, n)]
    elif kind == 'tiny_scale':
        x = (rng.random((n, d)) * 1e-20).astype(np.float32)
    else:
        x = rng.random((n, d), dtype=np.float32)
        x[:, ::2] *= 1e-6                                     # half the coordinates fall into the fp16 subnormal range
    want_i, want_d = ko.knn(x, k)
    for coarse, form in ((True, 1), (True, 2), (False, 0)):    # coarse tile with the norm K-step / norms through the accumulator; fp32 lists
        hipknn.use_coarse_collect(coarse)
        hipknn.set_collect_norm_form(form)
        try:
            idx, dist = cc.knn_graph(x, k)
        finally:
            hipknn.use_coarse_collect(True)
            hipknn.set_collect_norm_form(0)
        assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d), (kind, coarse, form)


@pytest.mark.parametrize('n,d,k,dup', [(9000, 64, 15, True), (20011, 64, 50, False), (4100, 50, 15, True), (70000, 64, 15, False)])
def test_collect_groups_never_change_the_result(hip, oracle_built, n, d, k, dup):
    """The coarse collect runs 32 or 64 queries per wave (one or two query groups sharing every candidate fragment; a
    four-group form at two waves per SIMD was built in round 4, measured 4-8 % slower and left as
    tools/experiments/knn_collect_qw4.patch).  Whatever the form, and whatever the
    number of candidate ranges (stages that straddle a range end, the padded end of the panel, the blocks that hold a
    wave's own queries), the lists are the oracle's.  n = 70000 is checked on sampled queries."""
    rng = np.random.default_rng(n + d + k)
    x = rng.random((n, d), dtype=np.float32)
    if dup:
        x[rng.choice(n, n // 20, replace=False)] = x[rng.choice(n, n // 20)]
    picks = np.arange(n) if n <= 21000 else np.concatenate([rng.choice(n, 60, replace=False), [0, 31, 32, n - 1]])
    want = [ko.knn(x, k, q_begin=int(q), n_q=1) for q in picks] if n > 21000 else None
    if want is None:
        want_i, want_d = ko.knn(x, k)
    try:
        for groups, splits in [(1, 0), (2, 0), (2, 3), (2, 7), (1, 16), (1, 5)]:
            hipknn.set_collect_groups(groups)
            hip.cc_knn_set_option(None, 5, splits)
            idx, dist = cc.knn_graph(x, k)
            if want is None:
                assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d), (groups, splits)
            else:
                for q, (wi, wd) in zip(picks, want):
                    assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0]), (groups, splits, int(q))
    finally:
        hipknn.set_collect_groups(0)
        hip.cc_knn_set_option(None, 5, 0)


@pytest.mark.parametrize('n,d,k,dup', [(9000, 64, 15, True), (20011, 64, 50, False), (4100, 50, 15, True), (6007, 32, 15, True),
                                       (5003, 16, 15, False), (3001, 3, 15, True), (70000, 64, 15, False)])
def test_collect_norm_forms_never_change_the_result(hip, oracle_built, n, d, k, dup):
    """The coarse collect's tile takes the two norms either in a K-step of their own (ND + 1 MFMAs, the pass test is the
    accumulator's sign) or through the accumulator (ND MFMAs: C = the candidates' scaled norms, a pair passes below a
    per-lane threshold; the library picks it when hits are rare).  Forced either way — every latent width's kernel
    (ND = 1, 2, 4), one or two query groups, ranges that end inside a stage, the partly filled last block, own-query
    blocks, duplicates — the lists are the oracle's.  n = 70000 is checked on sampled queries."""
    rng = np.random.default_rng(n + 3 * d + k)
    x = rng.random((n, d), dtype=np.float32)
    if dup:
        x[rng.choice(n, n // 20, replace=False)] = x[rng.choice(n, n // 20)]
    picks = np.arange(n) if n <= 21000 else np.concatenate([rng.choice(n, 60, replace=False), [0, 31, 32, n - 1]])
    want = [ko.knn(x, k, q_begin=int(q), n_q=1) for q in picks] if n > 21000 else None
    if want is None:
        want_i, want_d = ko.knn(x, k)
    try:
        for form, groups, splits in [(2, 0, 0), (2, 1, 0), (2, 2, 3), (2, 2, 7), (2, 1, 16), (1, 0, 0), (0, 0, 0)]:
            hipknn.set_collect_norm_form(form)
            hipknn.set_collect_groups(groups)
            hip.cc_knn_set_option(None, 5, splits)
            idx, dist = cc.knn_graph(x, k)
            ran = hipknn.last_collect_norm_form()
            assert ran == form if form else ran in (1, 2), (form, ran)      # the forced form is the one that ran
            if want is None:
                assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d), (form, groups, splits)
            else:
                for q, (wi, wd) in zip(picks, want):
                    assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0]), (form, groups, splits, int(q))
    finally:
        hipknn.set_collect_norm_form(0)
        hipknn.set_collect_groups(0)
        hip.cc_knn_set_option(None, 5, 0)


@pytest.mark.parametrize('n,d,k', [(600, 16, 15), (5000, 64, 15), (5000, 33, 50), (7000, 64, 128), (3000, 7, 100)])
def test_sample_layouts_never_change_the_result(hip, oracle_built, n, d, k):
    """The threshold of stage 1b is the m-th smallest of the per-slot minima of the sampled tiles: 64 slots per query
    (128 for m > 20), times the candidate ranges the pass is split over.  Whatever the stride, the number of ranges
    or the engine (fp32 list select with the option off), the neighbour lists are the oracle's; a looser or tighter
    threshold only moves work between the stages."""
    rng = np.random.default_rng(n * 7 + d + k)
    x = rng.random((n, d), dtype=np.float32)
    x[rng.choice(n, n // 25, replace=False)] = x[rng.choice(n, n // 25)]
    want_i, want_d = ko.knn(x, k)
    try:
        for splits, stride in [(0, 0), (1, 8), (3, 8), (8, 8), (16, 1), (2, 2), (1, 5), (4, 16)]:
            hipknn.set_sample_layout(splits, stride)
            idx, dist = cc.knn_graph(x, k)
            assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d), (splits, stride)
        hipknn.set_sample_layout(0, 0)
        for groups in (1, 2):                                # 32 or 64 queries per wave of the sample pass
            hipknn.set_sample_layout(0, 0, groups)
            idx, dist = cc.knn_graph(x, k)
            assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d), groups
        hipknn.set_sample_layout(0, 0)
        idx0, _ = cc.knn_graph(x, k)
        fb_slots = hipknn.last_fallback_count()
        hipknn.use_fp16_sample(False)
        idx, dist = cc.knn_graph(x, k)
        assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d)
        # the slot-minima threshold is looser than the exact m-th best of the sample, never tighter: it cannot send
        # more queries to the float64 stage for lack of candidates
        assert fb_slots <= hipknn.last_fallback_count() + n // 100
    finally:
        hipknn.set_sample_layout(0, 0)
        hipknn.use_fp16_sample(True)


def test_near_ties_below_fp32_resolution(hip, oracle_built):
    # neighbours whose distances differ by far less than the fp32 error of stage 1
    rng = np.random.default_rng(6)
    base = rng.random((1, 64), dtype=np.float32) * 100
    x = np.repeat(base, 600, axis=0)
    x[:, 0] += (np.arange(600) * 1e-5).astype(np.float32)
    x = np.concatenate([x, rng.random((600, 64), dtype=np.float32) * 100])
    check(x, 15)


def test_query_subrange_on_device(hip, oracle_built):
    import torch
    x = np.random.default_rng(12).random((3000, 64), dtype=np.float32)
    t = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    idx, dist = hipknn.knn_graph_device(t.data_ptr(), 3000, 64, 15, q_begin=1111, n_q=777)
    want_i, want_d = ko.knn(x, 15, q_begin=1111, n_q=777)
    assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d)


def test_results_into_pageable_pinned_and_unaligned_arrays(hip, oracle_built):
    """Pageable result arrays are populated under the kernels and filled through the pinned bounce buffer; pinned ones
    take the DMA directly.  Same bits either way, for arrays that start in the middle of a page too, and not a byte
    outside the arrays changes (the populate works on whole pages around them)."""
    import torch
    n, d, k = 9000, 64, 15
    x = np.random.default_rng(77).random((n, d), dtype=np.float32)
    t = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    want_i, want_d = ko.knn(x, k)
    idx, dist = hipknn.knn_graph_device(t.data_ptr(), n, d, k)                       # fresh pageable arrays
    assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d)
    pin_i = torch.empty((n, k), dtype=torch.int32).pin_memory()
    pin_d = torch.empty((n, k), dtype=torch.float32).pin_memory()
    hipknn.knn_graph_device(t.data_ptr(), n, d, k, out=(pin_i.numpy(), pin_d.numpy()))
    assert np.array_equal(pin_i.numpy(), want_i) and np.array_equal(pin_d.numpy(), want_d)
    words, lead = n * k, 1237                                                       # 4948 bytes into the buffer
    buf_i = np.full(words + 3000, -7, dtype=np.int32)
    buf_d = np.full(words + 3000, -7.0, dtype=np.float32)
    out_i, out_d = buf_i[lead:lead + words].reshape(n, k), buf_d[lead:lead + words].reshape(n, k)
    hipknn.knn_graph_device(t.data_ptr(), n, d, k, out=(out_i, out_d))
    assert np.array_equal(out_i, want_i) and np.array_equal(out_d, want_d)
    for buf, fill in ((buf_i, -7), (buf_d, -7.0)):
        assert np.all(buf[:lead] == fill) and np.all(buf[lead + words:] == fill)
    idx, dist = hipknn.knn_graph_device(t.data_ptr(), n, d, k, q_begin=8999, n_q=1)  # a 60-byte result
    assert np.array_equal(idx, want_i[8999:]) and np.array_equal(dist, want_d[8999:])


def test_sample_keys_hold_more_tiles_than_one_range(hip, oracle_built):
    """A sample key carries the index of its tile in 13 bits: a candidate range of more than 8192 sampled tiles is split
    (here: 300k candidates with EVERY block sampled = 9375 tiles).  512 queries from the middle, against the oracle."""
    import torch
    n, d, k = 300000, 8, 15
    x = np.random.default_rng(99).random((n, d), dtype=np.float32)
    t = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    want_i, want_d = ko.knn(x, k, q_begin=150000, n_q=512)
    try:
        for splits, stride in [(0, 1), (1, 1), (0, 2), (0, 0)]:
            hipknn.set_sample_layout(splits, stride)
            idx, dist = hipknn.knn_graph_device(t.data_ptr(), n, d, k, q_begin=150000, n_q=512)
            assert np.array_equal(idx, want_i) and np.array_equal(dist, want_d), (splits, stride)
            assert hipknn.last_fallback_count() < 64
    finally:
        hipknn.set_sample_layout(0, 0)


@pytest.mark.parametrize('n,d,k,include_self', [(20000, 64, 15, False), (12000, 3, 15, True), (30000, 32, 50, False),
                                                  (9000, 8, 1, False), (16411, 17, 16, False)])
def test_sampled_threshold_route(hip, oracle_built, n, d, k, include_self):
    # large candidate sets take stage 1b (threshold from a sample of the candidate blocks, then collect); the same
    # inputs through the full select (stage 1a) must give the same bits
    rng = np.random.default_rng(n + d + k)
    x = rng.random((n, d), dtype=np.float32)
    x[rng.choice(n, n // 20, replace=False)] = x[rng.choice(n, n // 20)]        # 5 % exact duplicates
    qs = rng.choice(n, 300, replace=False)
    idx, dist = cc.knn_graph(x, k, include_self)
    n_fb = hipknn.last_fallback_count()
    assert 3 <= hipknn.last_sample_stride() <= 8                                # route 1b ran, with a sample pass
    for q in qs:
        wi, wd = ko.knn(x, k, include_self, q_begin=int(q), n_q=1)
        assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0])
    hipknn.force_full_select(True)
    try:
        idx_a, dist_a = cc.knn_graph(x, k, include_self)
        assert hipknn.last_sample_stride() == 0                                 # route 1a: no sample
    finally:
        hipknn.force_full_select(False)
    assert np.array_equal(idx, idx_a) and np.array_equal(dist, dist_a)
    hipknn.use_fp16_collect(False)                                            # collect pass on the fp32 matrix cores
    try:
        idx_b, dist_b = cc.knn_graph(x, k, include_self)
    finally:
        hipknn.use_fp16_collect(True)
    assert np.array_equal(idx, idx_b) and np.array_equal(dist, dist_b)
    assert n_fb < n // 50                                                       # the threshold rarely misses


def test_sampled_route_clustered_and_sorted_data(hip, oracle_built):
    # cells sorted by cluster (index-correlated neighbours) and tight clusters far from the origin
    rng = np.random.default_rng(99)
    centres = rng.random((40, 16), dtype=np.float32) * 50 + 100
    lab = np.sort(rng.integers(0, 40, 24000))
    x = (centres[lab] + rng.standard_normal((24000, 16)).astype(np.float32) * 0.01).astype(np.float32)
    idx, dist = cc.knn_graph(x, 15)
    for q in rng.choice(24000, 200, replace=False):
        wi, wd = ko.knn(x, 15, q_begin=int(q), n_q=1)
        assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0])


def test_k_too_large_is_refused(hip):
    x = np.random.default_rng(0).random((10, 4), dtype=np.float32)
    with pytest.raises(ValueError):
        cc.knn_graph(x, 10)
    assert cc.knn_graph(x, 10, include_self=True)[0].shape == (10, 10)
    with pytest.raises(ValueError):
        cc.knn_graph(x.ravel(), 2)


def test_full_size_sampled_queries(hip, oracle_built):
    # BASELINE config 3 size (100k x 64, k=15): every query is computed on the GPU, a seeded
    # sample of 256 queries is checked bit-exact against the oracle.
    rng = np.random.default_rng(2024)
    x = (1.0 / (1.0 + np.exp(-rng.standard_normal((100000, 64))))).astype(np.float32)   # sigmoid-shaped
    idx, dist = cc.knn_graph(x, 15)
    qs = rng.choice(100000, 256, replace=False)
    for q in qs:
        wi, wd = ko.knn(x, 15, q_begin=int(q), n_q=1)
        assert np.array_equal(idx[q], wi[0]) and np.array_equal(dist[q], wd[0])
    # size-independent properties on all rows
    assert np.all(idx >= 0) and np.all(idx < 100000)
    assert np.all(idx != np.arange(100000)[:, None])
    assert np.all(np.diff(dist, axis=1) >= 0)
