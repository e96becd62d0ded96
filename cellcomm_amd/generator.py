"""HipGenerator — what a ``generator_factory`` returns in place of the Keras generator Model.

``BasicBiGan.generate_cells`` (reference src/bigan_basic.py:39-44) only calls
``generator.predict((encoding_in, random_in))`` and rounds the result; this object offers that
plus Keras-order ``get_weights``/``set_weights`` and ``layers`` and runs the forward pass on the
GPU (csrc/generator.hip).  Graphs: reference src/bigan_cont.py:7-25 (``cont``) and
src/bigan_classify.py:10-25 (``cls``).  It is built only on request (the factories below): at
G = 30000 the cont generator holds 0.22·G² = 198M weights.
"""
import ctypes

import numpy as np

from . import _lib
from .encoder import ARCH_CONT, ARCH_CLS, Layer, _ARCH_CODE


def _layer_plan(arch, encoding_size, gene_size):
    if arch == ARCH_CONT:
        w0, w1 = int(gene_size * 0.2), int(gene_size * 0.1)      # reference src/bigan_cont.py:8
        return [('gen_encoding_in', 'InputLayer', None, (), encoding_size),
                ('gen_random_in', 'InputLayer', None, (), encoding_size),
                ('concatenate', 'Concatenate', None, (), 2 * encoding_size),
                ('dense', 'Dense', 'sigmoid', (0, 1), 50),
                ('concatenate_1', 'Concatenate', None, (), 50 + 2 * encoding_size),
                ('dropout', 'Dropout', None, (), None),
                ('dense_1', 'Dense', 'sigmoid', (2, 3), 256),
                ('dense_2', 'Dense', 'sigmoid', (4, 5), 256),
                ('batch_normalization', 'BatchNormalization', None, (6, 7, 8, 9), 256),
                ('concatenate_2', 'Concatenate', None, (), 256 + 2 * encoding_size),
                ('dropout_1', 'Dropout', None, (), None),
                ('dense_3', 'Dense', 'sigmoid', (10, 11), w1),
                ('dense_4', 'Dense', 'relu', (12, 13), w0),
                ('batch_normalization_1', 'BatchNormalization', None, (14, 15, 16, 17), w0),
                ('gen_encoding_out', 'Dense', 'relu', (18, 19), gene_size)]
    return [('gen_encoding_in', 'InputLayer', None, (), encoding_size),
            ('gen_random_in', 'InputLayer', None, (), encoding_size),
            ('concatenate', 'Concatenate', None, (), 2 * encoding_size),
            ('dense', 'Dense', 'sigmoid', (0, 1), 50),
            ('concatenate_1', 'Concatenate', None, (), 50 + 2 * encoding_size),
            ('dropout', 'Dropout', None, (), None),
            ('dense_1', 'Dense', 'sigmoid', (2, 3), 256),
            ('batch_normalization', 'BatchNormalization', None, (4, 5, 6, 7), 256),
            ('concatenate_2', 'Concatenate', None, (), 256 + 2 * encoding_size),
            ('dropout_1', 'Dropout', None, (), None),
            ('dense_2', 'Dense', 'sigmoid', (8, 9), 256),
            ('dropout_2', 'Dropout', None, (), None),
            ('dense_3', 'Dense', 'relu', (10, 11), 1024),
            ('dense_4', 'Dense', 'relu', (12, 13), gene_size)]


class HipGenerator:
    name = 'cell_generator'

    def __init__(self, encoding_size, gene_size, arch=ARCH_CONT, seed=None):
        if arch not in _ARCH_CODE:
            raise ValueError(f'unknown generator arch {arch!r}')
        self.arch, self.encoding_size, self.gene_size = arch, int(encoding_size), int(gene_size)
        self.output_shape = (None, self.gene_size)
        self.trainable = False
        lib = _lib.ensure_init()
        self._h = ctypes.c_void_p()
        _lib.check(lib.cc_generator_create(_ARCH_CODE[arch], self.gene_size, self.encoding_size, ctypes.byref(self._h)))
        self._shapes = []
        for i in range(lib.cc_generator_num_weights(self._h)):
            r, c = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(lib.cc_generator_weight_shape(self._h, i, ctypes.byref(r), ctypes.byref(c)))
            self._shapes.append((r.value, c.value) if c.value >= 0 else (r.value,))
        self._weights = None
        self.layers = [Layer(self, *spec) for spec in _layer_plan(arch, self.encoding_size, self.gene_size)]
        self.set_weights(self._initial_weights(seed))

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                _lib.lib().cc_generator_free(self._h)
        except Exception:
            pass

    def _initial_weights(self, seed):
        """Keras defaults: glorot_uniform kernels, zero biases, fresh BatchNormalization state."""
        rng = np.random.default_rng(seed)
        bn_first = {ARCH_CONT: (6, 14), ARCH_CLS: (4,)}[self.arch]
        bn = {}
        for f in bn_first:
            bn.update({f: 1.0, f + 1: 0.0, f + 2: 0.0, f + 3: 1.0})
        out = []
        for i, shp in enumerate(self._shapes):
            if len(shp) == 2:
                limit = np.sqrt(6.0 / max(shp[0] + shp[1], 1))
                out.append(rng.uniform(-limit, limit, size=shp).astype(np.float32))
            else:
                out.append(np.full(shp, bn.get(i, 0.0), dtype=np.float32))
        return out

    def get_weights(self):
        return [w.copy() for w in self._weights]

    def set_weights(self, weights):
        weights = list(weights)
        if len(weights) != len(self._shapes):
            raise ValueError(f'You called `set_weights(weights)` on layer "{self.name}" with a weight list of '
                             f'length {len(weights)}, but the layer was expecting {len(self._shapes)} weights.')
        arrs = []
        for w, shp in zip(weights, self._shapes):
            a = np.ascontiguousarray(w, dtype=np.float32)
            if a.shape != tuple(shp):
                raise ValueError(f'Layer weight shape {tuple(shp)} not compatible with provided weight shape {a.shape}')
            arrs.append(a)
        lib = _lib.lib()
        for i, a in enumerate(arrs):
            _lib.check(lib.cc_generator_set_weights(self._h, i, a.ctypes.data, a.shape[0],
                                                    a.shape[1] if a.ndim == 2 else -1))
        self._weights = arrs

    def count_params(self):
        return int(sum(w.size for w in self._weights))

    def predict(self, inputs, batch_size=None, verbose=0, **_ignored):
        """inputs = (encoding_in [n x E], random_in [n x E]) -> np.float32 [n x gene_size]."""
        enc_in, rand_in = inputs
        enc_in = np.asarray(enc_in, dtype=np.float32)
        rand_in = np.asarray(rand_in, dtype=np.float32)
        want = (len(enc_in), self.encoding_size)
        if enc_in.shape != want or rand_in.shape != want:
            raise ValueError(f'Input of layer "{self.name}" is incompatible with the layer: expected two arrays of '
                             f'shape (None, {self.encoding_size}), found {enc_in.shape} and {rand_in.shape}')
        x = np.ascontiguousarray(np.concatenate([enc_in, rand_in], axis=1))
        out = np.empty((len(x), self.gene_size), dtype=np.float32)
        _lib.check(_lib.ensure_init().cc_generate_dense(self._h, x.ctypes.data, len(x), out.ctypes.data))
        return out

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"  [{self.arch} generator on HIP/gfx950]')
        for lay in self.layers:
            print_fn(f'{lay.name + " (" + lay.kind + ")":<44}{lay.count_params():>12}')
        print_fn(f'Total params: {self.count_params():,}')


def build_cont_generator(encoding_size, gene_size):
    """generator_factory with the graph of reference src/bigan_cont.py:7-25."""
    return HipGenerator(encoding_size, gene_size, ARCH_CONT)


def build_classify_generator(encoding_size, gene_size):
    """generator_factory with the graph of reference src/bigan_classify.py:10-25."""
    return HipGenerator(encoding_size, gene_size, ARCH_CLS)
