# accumulator-form collect at FOUR waves per SIMD, made possible by the un-batched fp32 re-check (fine_dist: 8 instead of 64 registers
# of loads in flight): _v1 = one block at a time (109 registers), _v2 = two blocks at a time (128 registers, 20 B of scratch).
# Libraries built from patched copies of knn.hip: norm form 3 = the four-wave kernel, norm form 2 = the product's structure with the
# un-batched re-check (what that alone costs); "product" = the tree's library (batched re-check).
cd /root/repo
cp cellcomm_amd/libcellcomm_hip.so /tmp/cc_product.so
trap 'cp /tmp/cc_product.so cellcomm_amd/libcellcomm_hip.so' EXIT
for lib in tools/experiments/_v1.so tools/experiments/_v2.so; do
cp $lib cellcomm_amd/libcellcomm_hip.so
timeout -k 10 300 python - <<'PY' || exit 1
import numpy as np, cellcomm_amd as cc
from cellcomm_amd import knn as hipknn, _lib
rng = np.random.default_rng(5)
for n, k in ((70001, 15), (20011, 50), (9000, 15)):
    x = rng.random((n, 64), dtype=np.float32)
    x[rng.choice(n, n // 20, replace=False)] = x[rng.choice(n, n // 20)]
    hipknn.set_collect_groups(2)
    res = {}
    for form in (1, 2, 3):
        for splits in (0, 3, 7):
            _lib.check(_lib.lib().cc_knn_set_option(None, 11, form)); _lib.check(_lib.lib().cc_knn_set_option(None, 5, splits))
            res[form, splits] = cc.knn_graph(x, k)
    for key, (i, d) in res.items():
        assert np.array_equal(i, res[1, 0][0]) and np.array_equal(d, res[1, 0][1]), (n, k, key)
    print('parity ok', n, k, flush=True)
PY
done
run() { for shape in "1000000 64 15 125000 5" "500000 64 50 62500 10" "100000 64 15 100000 20"; do
  CC_KNN_COLLECT_NORM_FORM=$1 timeout -k 10 120 python tools/knn_bench.py $shape 2>&1 | grep "^n=" | cut -c1-140; done; }
for rep in 1 2; do
cp /tmp/cc_product.so cellcomm_amd/libcellcomm_hip.so; echo "== product form 2"; run 2
cp tools/experiments/_v1.so cellcomm_amd/libcellcomm_hip.so; echo "== v1 form 2 (3 waves, un-batched)"; run 2; echo "== v1 form 3 (4 waves, one block)"; run 3
cp tools/experiments/_v2.so cellcomm_amd/libcellcomm_hip.so; echo "== v2 form 3 (4 waves, two blocks)"; run 3
done
