"""Oracle: BiGAN encoder forward pass in NumPy (float64 truth / float32 Keras-like).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  PARITY UNPINNED vs TF: this
restates the Keras 2.4 layer definitions used by the reference graphs:

  cont  reference src/bigan_cont.py:28-41
        Dense(int(G*0.1), sigmoid) -> Dropout -> Concatenate([x, cell_in])
        -> Dense(int(G*0.05), sigmoid) -> Dropout -> BatchNormalization
        -> Dense(150, sigmoid) -> Dense(150, sigmoid) -> Dense(E, sigmoid)
  cls   reference src/bigan_classify.py:28-40
        p = Dense(1000, sigmoid)(cell_in) -> Dropout -> Dense(300, sigmoid)
        -> Dropout -> Concatenate([x, p]) -> Dense(150, sigmoid)
        -> Dense(E, softmax)

Keras semantics restated: Dense is ``x @ kernel + bias`` with kernel [in, out];
Dropout is the identity at inference; BatchNormalization at inference is
``(x - moving_mean) / sqrt(moving_variance + 1e-3) * gamma + beta`` (Keras default
epsilon=1e-3); Concatenate is along the last axis; ``Model.predict`` casts the
input to float32 and runs batches of 32 (reference src/bigan_basic.py:28-30).
"""
import numpy as np

ARCH_CONT = 'cont'
ARCH_CLS = 'cls'
BN_EPSILON = 1e-3


def hidden_widths(arch, gene_size):
    if arch == ARCH_CONT:
        # reference src/bigan_cont.py:29
        return [int(gene_size * f) for f in [0.1, 0.05]]
    return [1000, 300]


def weight_shapes(arch, encoding_size, gene_size):
    """Shapes of Model.get_weights() in Keras order."""
    g, e = gene_size, encoding_size
    w0, w1 = hidden_widths(arch, gene_size)
    if arch == ARCH_CONT:
        return [(g, w0), (w0,), (w0 + g, w1), (w1,), (w1,), (w1,), (w1,), (w1,),
                (w1, 150), (150,), (150, 150), (150,), (150, e), (e,)]
    return [(g, 1000), (1000,), (1000, 300), (300,), (1300, 150), (150,), (150, e), (e,)]


def glorot_weights(arch, encoding_size, gene_size, seed, randomize_bn=False, bias_scale=0.0):
    """Keras default initialisers (glorot_uniform kernels, zero biases, fresh
    BatchNormalization state) drawn from NumPy's default_rng(seed).  With
    randomize_bn / bias_scale the BN statistics and biases are made non-trivial
    so that fixtures exercise them."""
    rng = np.random.default_rng(seed)
    out = []
    shapes = weight_shapes(arch, encoding_size, gene_size)
    bn_slots = {4, 5, 6, 7} if arch == ARCH_CONT else set()
    for i, shp in enumerate(shapes):
        if len(shp) == 2:
            limit = np.sqrt(6.0 / max(shp[0] + shp[1], 1))
            out.append(rng.uniform(-limit, limit, size=shp).astype(np.float32))
        elif i in bn_slots:
            n = shp[0]
            if randomize_bn:
                v = {4: rng.uniform(0.5, 1.5, n), 5: rng.uniform(-0.5, 0.5, n),
                     6: rng.uniform(0.2, 0.8, n), 7: rng.uniform(0.05, 0.5, n)}[i]
            else:
                v = {4: np.ones(n), 5: np.zeros(n), 6: np.zeros(n), 7: np.ones(n)}[i]
            out.append(v.astype(np.float32))
        else:
            out.append((rng.uniform(-1, 1, shp) * bias_scale).astype(np.float32))
    return out


def _sigmoid(z):
    out = np.empty_like(z)
    pos = z >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-z[pos]))
    ez = np.exp(z[~pos])
    out[~pos] = ez / (1.0 + ez)
    return out


def _softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def forward(arch, weights, x, dtype=np.float64):
    """Encoder forward.  dtype=float64: truth; dtype=float32: what a float32
    framework computes up to summation order."""
    w = [np.asarray(a, dtype=dtype) for a in weights]
    x = np.asarray(x, dtype=np.float32).astype(dtype)  # predict() casts input to float32 first
    if arch == ARCH_CONT:
        k1, b1, k2, b2, gamma, beta, mean, var, k3, b3, k4, b4, k5, b5 = w
        h = _sigmoid(x @ k1 + b1)
        c = np.concatenate([h, x], axis=-1)          # hidden first: reference bigan_cont.py:34
        h = _sigmoid(c @ k2 + b2)
        h = (h - mean) / np.sqrt(var + dtype(BN_EPSILON)) * gamma + beta
        h = _sigmoid(h @ k3 + b3)
        h = _sigmoid(h @ k4 + b4)
        return _sigmoid(h @ k5 + b5)
    k1, b1, k2, b2, k3, b3, k4, b4 = w
    p = _sigmoid(x @ k1 + b1)
    h = _sigmoid(p @ k2 + b2)
    c = np.concatenate([h, p], axis=-1)              # h first: reference bigan_classify.py:37
    h = _sigmoid(c @ k3 + b3)
    return _softmax(h @ k4 + b4)


def predict_like_keras(arch, weights, x, batch_size=32):
    """float32 forward in batches of 32, concatenated — the shape of
    keras Model.predict as called at reference src/bigan_basic.py:30."""
    x = np.asarray(x, dtype=np.float32)
    outs = [forward(arch, weights, x[i:i + batch_size], dtype=np.float32)
            for i in range(0, len(x), batch_size)]
    if not outs:
        return np.zeros((0, np.asarray(weights[-1]).shape[0]), dtype=np.float32)
    return np.concatenate(outs, axis=0).astype(np.float32)


def trainings_encoding_prediction_cls(prediction, encoding_size):
    """argmax(-1) -> one-hot float32; reference src/bigan_classify.py:121-124."""
    prediction = np.asarray(prediction)
    am = prediction.argmax(-1)
    out = np.zeros((len(am), encoding_size), dtype=np.float32)
    out[np.arange(len(am)), am] = 1.0
    return out


def _pow2_scale(w):
    """The power-of-two scale the fp16 paths apply to a weight matrix before rounding it to
    fp16 (cellcomm_amd/csrc/gemm16.hip pack_weight16): max|w|*s lands in [2^13, 2^14)."""
    mx = float(np.max(np.abs(w))) if w.size else 0.0
    if mx == 0.0 or not np.isfinite(mx):
        return 1.0
    _, e = np.frexp(np.float32(mx))
    return float(2.0 ** (14 - int(e)))


def _q16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float64)


def forward_f16(arch, weights, x):
    """Float64 evaluation of what CC_DTYPE_F16 is defined to compute: the raw input, the weights of
    the first two layers (after their power-of-two scale) and, for the cont arch, the first hidden
    activation are each rounded once to fp16; everything else as forward().  BASELINE config 5
    ("fp16 MFMA") is checked against this."""
    w = [np.asarray(a, dtype=np.float64) for a in weights]
    xq = _q16(x)
    if arch == ARCH_CONT:
        k1, b1, k2, b2, gamma, beta, mean, var, k3, b3, k4, b4, k5, b5 = w
        s1, s2 = _pow2_scale(k1), _pow2_scale(k2)
        k1q, k2q = _q16(k1 * s1) / s1, _q16(k2 * s2) / s2
        h = _q16(_sigmoid(xq @ k1q + b1))
        c = np.concatenate([h, xq], axis=-1)
        h = _sigmoid(c @ k2q + b2)
        h = (h - mean) / np.sqrt(var + BN_EPSILON) * gamma + beta
        h = _sigmoid(h @ k3 + b3)
        h = _sigmoid(h @ k4 + b4)
        return _sigmoid(h @ k5 + b5)
    k1, b1, k2, b2, k3, b3, k4, b4 = w
    s1 = _pow2_scale(k1)
    p = _sigmoid(xq @ (_q16(k1 * s1) / s1) + b1)
    h = _sigmoid(p @ k2 + b2)
    c = np.concatenate([h, p], axis=-1)
    h = _sigmoid(c @ k3 + b3)
    return _softmax(h @ k4 + b4)


# ---- generator (reference src/bigan_cont.py:7-25, src/bigan_classify.py:10-25) -----------------

def generator_weight_shapes(arch, encoding_size, gene_size):
    """Shapes of the generator's Model.get_weights() in Keras order."""
    g, i = gene_size, 2 * encoding_size
    if arch == ARCH_CONT:
        w0, w1 = int(g * 0.2), int(g * 0.1)
        return [(i, 50), (50,), (50 + i, 256), (256,), (256, 256), (256,), (256,), (256,), (256,), (256,),
                (256 + i, w1), (w1,), (w1, w0), (w0,), (w0,), (w0,), (w0,), (w0,), (w0, g), (g,)]
    return [(i, 50), (50,), (50 + i, 256), (256,), (256,), (256,), (256,), (256,),
            (256 + i, 256), (256,), (256, 1024), (1024,), (1024, g), (g,)]


def generator_weights(arch, encoding_size, gene_size, seed, randomize_bn=True, bias_scale=0.1):
    rng = np.random.default_rng(seed)
    shapes = generator_weight_shapes(arch, encoding_size, gene_size)
    bn_first = (6, 14) if arch == ARCH_CONT else (4,)
    out = []
    for idx, shp in enumerate(shapes):
        if len(shp) == 2:
            limit = np.sqrt(6.0 / max(shp[0] + shp[1], 1))
            out.append(rng.uniform(-limit, limit, size=shp).astype(np.float32))
            continue
        role = next((idx - f for f in bn_first if 0 <= idx - f < 4), None)
        n = shp[0]
        if role is None:
            out.append((rng.uniform(-1, 1, n) * bias_scale).astype(np.float32))
        elif randomize_bn:
            out.append([rng.uniform(0.5, 1.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(0.2, 0.8, n),
                        rng.uniform(0.05, 0.5, n)][role].astype(np.float32))
        else:
            out.append([np.ones(n), np.zeros(n), np.zeros(n), np.ones(n)][role].astype(np.float32))
    return out


def _bn(h, gamma, beta, mean, var, dtype):
    return (h - mean) / np.sqrt(var + dtype(BN_EPSILON)) * gamma + beta


def generator_forward(arch, weights, encoding_in, random_in, dtype=np.float64):
    """Generator forward before the rounding of generate_cells (reference src/bigan_basic.py:39-44)."""
    w = [np.asarray(a, dtype=dtype) for a in weights]
    all_in = np.concatenate([np.asarray(encoding_in, dtype=np.float32), np.asarray(random_in, dtype=np.float32)],
                            axis=-1).astype(dtype)
    relu = lambda z: np.maximum(z, 0)
    if arch == ARCH_CONT:
        k1, b1, k2, b2, k3, b3, g1, be1, m1, v1, k4, b4, k5, b5, g2, be2, m2, v2, k6, b6 = w
        x = _sigmoid(all_in @ k1 + b1)
        x = _sigmoid(np.concatenate([x, all_in], -1) @ k2 + b2)
        x = _sigmoid(x @ k3 + b3)
        x = _bn(x, g1, be1, m1, v1, dtype)
        x = _sigmoid(np.concatenate([x, all_in], -1) @ k4 + b4)
        x = relu(x @ k5 + b5)
        x = _bn(x, g2, be2, m2, v2, dtype)
        return relu(x @ k6 + b6)
    k1, b1, k2, b2, g1, be1, m1, v1, k3, b3, k4, b4, k5, b5 = w
    x = _sigmoid(all_in @ k1 + b1)
    x = _sigmoid(np.concatenate([x, all_in], -1) @ k2 + b2)
    x = _bn(x, g1, be1, m1, v1, dtype)
    x = _sigmoid(np.concatenate([x, all_in], -1) @ k3 + b3)
    x = relu(x @ k4 + b4)
    return relu(x @ k5 + b5)
