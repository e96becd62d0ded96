#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (run in the build container).

Encoder fixtures: seeded Glorot weights with non-trivial BatchNormalization
statistics and biases (regenerated from the seed by the tests; a checksum is
stored to detect RNG drift), seeded sparse integer counts as input, outputs of
the float64 forward (truth) and of the float32 Keras-like forward.
kNN fixtures: seeded uniform encodings with >= 5 % exact duplicate rows, indices
and distances from oracle/knn_oracle.c.

The reference itself cannot produce these (tensorflow==2.4.0 is not installable
here and the reference has no kNN), so they pin the ORACLE, not TensorFlow — see
DESIGN.md §Oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import encoder_oracle as eo  # noqa: E402
from oracle import knn_oracle as ko      # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')

ENCODER_CASES = [  # (arch, G, E, N)
    ('cont', 5, 8, 5), ('cont', 100, 12, 64), ('cont', 2000, 32, 256),
    ('cls', 100, 12, 64), ('cls', 2000, 32, 256),
]
KNN_CASES = [(257, 3), (2048, 32), (2048, 64)]  # (N, E); k in {15, 50}


def sparse_counts(rng, n, g, density=0.05):
    x = np.zeros((n, g), dtype=np.float32)
    mask = rng.random((n, g)) < max(density, 2.0 / g)
    x[mask] = np.minimum(rng.geometric(0.5, size=int(mask.sum())), 2047)
    return x


def weights_checksum(ws):
    return float(sum(np.abs(w.astype(np.float64)).sum() for w in ws))


def main():
    os.makedirs(GOLD, exist_ok=True)
    for arch, g, e, n in ENCODER_CASES:
        seed = 1000 + g + e
        ws = eo.glorot_weights(arch, e, g, seed, randomize_bn=True, bias_scale=0.1)
        x = sparse_counts(np.random.default_rng(seed + 1), n, g)
        out64 = eo.forward(arch, ws, x, np.float64)
        out32 = eo.predict_like_keras(arch, ws, x)
        np.savez_compressed(os.path.join(GOLD, f'encoder_{arch}_G{g}_E{e}_N{n}.npz'), arch=arch, G=g, E=e,
                            seed=seed, x=x, out64=out64, out32=out32, wsum=weights_checksum(ws))
        rel = np.abs(out32 - out64).max() / np.abs(out64).max()
        print(f'encoder {arch} G={g} E={e} N={n}: fp32-vs-fp64 max rel {rel:.2e}')
    for n, e in KNN_CASES:
        rng = np.random.default_rng(7000 + n + e)
        x = rng.random((n, e), dtype=np.float32)
        dup = rng.choice(n, size=max(n // 16, 8), replace=False)      # > 5 % rows become copies
        src = rng.choice(n, size=len(dup))
        x[dup] = x[src]
        x[:6] = x[0]                                                  # a 6-fold duplicate cluster
        out = {'x': x}
        for k in (15, 50):
            idx, dist = ko.knn(x, k)
            out[f'idx{k}'], out[f'dist{k}'] = idx, dist
        np.savez_compressed(os.path.join(GOLD, f'knn_N{n}_E{e}.npz'), **out)
        print(f'knn N={n} E={e} written')


if __name__ == '__main__':
    main()
