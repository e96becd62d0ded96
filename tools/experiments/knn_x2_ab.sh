# accumulator-form collect with stages of FOUR blocks in a ring of three (one barrier, one counted wait per four blocks; 54 KiB of
# LDS per workgroup: still three per CU) against the product's two-block stages in a ring of four.  Library built from a patched
# knn.hip (norm form 3 selects it); parity first.
cd /root/repo
cp cellcomm_amd/libcellcomm_hip.so /tmp/cc_product.so
trap 'cp /tmp/cc_product.so cellcomm_amd/libcellcomm_hip.so' EXIT
cp tools/experiments/_x2.so cellcomm_amd/libcellcomm_hip.so
timeout -k 10 300 python - <<'PY' || exit 1
import numpy as np, cellcomm_amd as cc
from cellcomm_amd import knn as hipknn, _lib
rng = np.random.default_rng(5)
for n, k in ((70001, 15), (20011, 50), (9000, 15)):
    x = rng.random((n, 64), dtype=np.float32)
    x[rng.choice(n, n // 20, replace=False)] = x[rng.choice(n, n // 20)]
    hipknn.set_collect_groups(2)
    res = {}
    for form in (1, 3):
        for splits in (0, 3, 7):
            _lib.check(_lib.lib().cc_knn_set_option(None, 11, form)); _lib.check(_lib.lib().cc_knn_set_option(None, 5, splits))
            res[form, splits] = cc.knn_graph(x, k)
    for key, (i, d) in res.items():
        assert np.array_equal(i, res[1, 0][0]) and np.array_equal(d, res[1, 0][1]), (n, k, key)
    print('parity ok', n, k, flush=True)
PY
for rep in 1 2; do for form in 2 3; do echo "== form $form"; for shape in "1000000 64 15 125000 5" "500000 64 50 62500 10"; do
  CC_KNN_COLLECT_NORM_FORM=$form timeout -k 10 120 python tools/knn_bench.py $shape 2>&1 | grep "^n=" | cut -c1-140; done; done; done
