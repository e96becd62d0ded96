"""HipEncoder — the object an ``encoder_factory`` returns in place of a Keras Model.

Drop-in boundary (reference src/bigan_basic.py:11-30): ``BasicBiGan`` only ever
calls ``encoder.predict(cell_data)``, ``encoder.summary()`` and iterates
``encoder.layers`` / ``layer.get_weights()``.  This class offers exactly that
surface (plus ``get_weights``/``set_weights`` in Keras order, ``input_shape``,
``output_shape``) and runs the forward pass on the GPU through the C ABI.

Graphs: reference src/bigan_cont.py:28-41 (``cont``) and
src/bigan_classify.py:28-40 (``cls``).
"""
import ctypes

import numpy as np

from . import _lib
from .matrix import CellMatrix

ARCH_CONT, ARCH_CLS = 'cont', 'cls'
_ARCH_CODE = {ARCH_CONT: 0, ARCH_CLS: 1}
# arithmetic of the two big layers: exact fp32 MFMA (default) | fp16 operands | fp16 hi+lo split operands
_DTYPE_CODE = {'f32': 0, 'f16': 1, 'f16x2': 2}


def hidden_widths(arch, gene_size):
    if arch == ARCH_CONT:
        return [int(gene_size * f) for f in [0.1, 0.05]]   # reference src/bigan_cont.py:29
    return [1000, 300]                                      # reference src/bigan_classify.py:31,34


class Layer:
    """Minimal stand-in for a Keras layer: a name, an activation name and the
    slice of the encoder's weight list it owns."""

    def __init__(self, encoder, name, kind, activation=None, weight_slots=(), units=None):
        self._encoder = encoder
        self.name = name
        self.kind = kind
        self.activation = activation
        self.units = units
        self._slots = tuple(weight_slots)

    def get_weights(self):
        return [self._encoder._weights[i].copy() for i in self._slots]

    def count_params(self):
        return int(sum(self._encoder._weights[i].size for i in self._slots))


def _layer_plan(arch, encoding_size, gene_size):
    """(name, kind, activation, weight slots, units) per Keras layer, in graph order."""
    w0, w1 = hidden_widths(arch, gene_size)
    if arch == ARCH_CONT:
        return [('enc_cell_in', 'InputLayer', None, (), gene_size),
                ('dense', 'Dense', 'sigmoid', (0, 1), w0),
                ('dropout', 'Dropout', None, (), None),
                ('concatenate', 'Concatenate', None, (), w0 + gene_size),
                ('dense_1', 'Dense', 'sigmoid', (2, 3), w1),
                ('dropout_1', 'Dropout', None, (), None),
                ('batch_normalization', 'BatchNormalization', None, (4, 5, 6, 7), w1),
                ('dense_2', 'Dense', 'sigmoid', (8, 9), 150),
                ('dense_3', 'Dense', 'sigmoid', (10, 11), 150),
                ('enc_encoding_out', 'Dense', 'sigmoid', (12, 13), encoding_size)]
    return [('enc_cell_in', 'InputLayer', None, (), gene_size),
            ('dense', 'Dense', 'sigmoid', (0, 1), 1000),
            ('dropout', 'Dropout', None, (), None),
            ('dense_1', 'Dense', 'sigmoid', (2, 3), 300),
            ('dropout_1', 'Dropout', None, (), None),
            ('concatenate', 'Concatenate', None, (), 1300),
            ('dense_2', 'Dense', 'sigmoid', (4, 5), 150),
            ('dense_3', 'Dense', 'softmax', (6, 7), encoding_size)]


class HipEncoder:
    name = 'cell_encoder'

    def __init__(self, encoding_size, gene_size, arch=ARCH_CONT, seed=None, batch_rows=None, dtype='f32', sparse=False,
                 low_latency=False):
        if arch not in _ARCH_CODE:
            raise ValueError(f'unknown encoder arch {arch!r}')
        if dtype not in _DTYPE_CODE:
            raise ValueError(f'unknown encoder dtype {dtype!r} (f32, f16 or f16x2)')
        self.dtype = dtype
        self.arch = arch
        self.encoding_size = int(encoding_size)
        self.gene_size = int(gene_size)
        self.input_shape = (None, self.gene_size)
        self.output_shape = (None, self.encoding_size)
        self.trainable = False
        lib = _lib.ensure_init()
        self._h = ctypes.c_void_p()
        _lib.check(lib.cc_encoder_create(_ARCH_CODE[arch], self.gene_size, self.encoding_size, _DTYPE_CODE[dtype],
                                         ctypes.byref(self._h)))
        if batch_rows:
            _lib.check(lib.cc_encoder_set_batch(self._h, int(batch_rows)))
        # sparse=True: CellMatrix rows in storage order never get densified — the products with the raw counts run
        # over the non-zeros only (fp32, csrc/spmm.hip); sampled rows and dense input keep the gather + dense path
        self.sparse = bool(sparse)
        if self.sparse:
            if dtype != 'f32':
                raise ValueError('sparse=True computes in fp32: use dtype="f32"')
            _lib.check(lib.cc_encoder_set_sparse_input(self._h, 1))
        # low_latency=True: calls of <= 128 rows (keras predict batches of 32, minibatches of 128) run the long-K
        # fp32 layers split-K over the whole chip (and the narrow tail in one launch); results stay within the same
        # tolerance and are reproducible, but are no longer bit-identical to the same rows inside a larger call
        self.low_latency = bool(low_latency)
        if self.low_latency:
            _lib.check(lib.cc_encoder_set_low_latency(self._h, 1))
        self._shapes = []
        for i in range(lib.cc_encoder_num_weights(self._h)):
            r, c = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(lib.cc_encoder_weight_shape(self._h, i, ctypes.byref(r), ctypes.byref(c)))
            self._shapes.append((r.value, c.value) if c.value >= 0 else (r.value,))
        self._weights = None
        self.layers = [Layer(self, *spec) for spec in _layer_plan(arch, self.encoding_size, self.gene_size)]
        self.set_weights(self._initial_weights(seed))

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                _lib.lib().cc_encoder_free(self._h)
        except Exception:
            pass

    # -- weights --------------------------------------------------------------------
    def _initial_weights(self, seed):
        """Keras defaults: glorot_uniform kernels, zero biases, BatchNormalization
        gamma=1, beta=0, moving_mean=0, moving_variance=1."""
        rng = np.random.default_rng(seed)
        bn = {4: 1.0, 5: 0.0, 6: 0.0, 7: 1.0} if self.arch == ARCH_CONT else {}
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
        for i, (w, shp) in enumerate(zip(weights, self._shapes)):
            a = np.ascontiguousarray(w, dtype=np.float32)
            if a.shape != tuple(shp):
                raise ValueError(f'Layer weight shape {tuple(shp)} not compatible with provided weight shape {a.shape}')
            arrs.append(a)
        lib = _lib.lib()
        for i, a in enumerate(arrs):
            rows = a.shape[0]
            cols = a.shape[1] if a.ndim == 2 else -1
            _lib.check(lib.cc_encoder_set_weights(self._h, i, a.ctypes.data, rows, cols))
        self._weights = arrs

    def count_params(self):
        return int(sum(w.size for w in self._weights))

    # -- forward ----------------------------------------------------------------------
    def predict(self, x, batch_size=None, verbose=0, **_ignored):
        """Encodings for `x`: CellMatrix (CSR rows, gathered on the GPU), scipy
        sparse matrix, or anything array-like [n x gene_size] (DataFrame, ndarray,
        list of lists) — returns np.float32 [n x encoding_size] like
        keras Model.predict (reference src/bigan_basic.py:30).  `batch_size` is
        accepted for signature compatibility; batching is internal."""
        lib = _lib.ensure_init()
        if hasattr(x, 'tocsr') and not isinstance(x, CellMatrix):
            x = CellMatrix.from_scipy(x)
        if isinstance(x, CellMatrix):
            self._check_width(x.shape[1])
            n = x.shape[0]
            out = np.empty((n, self.encoding_size), dtype=np.float32)
            ids = x.row_ids()
            ids_ptr = None
            if ids is not None:
                ids = np.ascontiguousarray(ids, dtype=np.int64)
                ids_ptr = ids.ctypes.data
            _lib.check(lib.cc_encode(self._h, x.device_handle(), ids_ptr, 0, n, out.ctypes.data))
            return out
        arr = np.asarray(x)
        if arr.ndim != 2:
            raise ValueError(f'Input 0 of layer "{self.name}" is incompatible with the layer: '
                             f'expected shape=(None, {self.gene_size}), found shape={arr.shape}')
        self._check_width(arr.shape[1])
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        out = np.empty((arr.shape[0], self.encoding_size), dtype=np.float32)
        _lib.check(lib.cc_encode_dense(self._h, arr.ctypes.data, arr.shape[0], out.ctypes.data))
        return out

    def predict_to_device(self, x, dev_out_ptr, row_start=0, n=None):
        """Encode CSR rows [row_start, row_start+n) of CellMatrix `x` into device
        memory at `dev_out_ptr` ([n x encoding_size] float32, contiguous)."""
        lib = _lib.ensure_init()
        self._check_width(x.shape[1])
        n = x.shape[0] - row_start if n is None else n
        ids = x.row_ids()
        ids_ptr = None
        if ids is not None:
            ids = np.ascontiguousarray(ids[row_start:row_start + n], dtype=np.int64)
            ids_ptr, row_start = ids.ctypes.data, 0
        _lib.check(lib.cc_encode_dev(self._h, x.device_handle(), ids_ptr, row_start, n, dev_out_ptr))

    @property
    def pass_rows(self):
        """Rows per pass through the layer stack (chosen from the layer widths unless `batch_rows` was given)."""
        return int(_lib.lib().cc_encoder_batch(self._h))

    def fp32_fallback_passes(self):
        """dtype 'f16x2': passes of the last predict whose input was not fp16-exact (ran in fp32)."""
        return int(_lib.lib().cc_encoder_fp32_fallback_passes(self._h))

    def __call__(self, x, training=False):
        return self.predict(x)

    def _check_width(self, g):
        if g != self.gene_size:
            raise ValueError(f'Input 0 of layer "{self.name}" is incompatible with the layer: '
                             f'expected shape=(None, {self.gene_size}), found shape=(None, {g})')

    # -- reporting ----------------------------------------------------------------------
    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"  [{self.arch} encoder on HIP/gfx950]')
        print_fn('_' * 65)
        print_fn(f'{"Layer (type)":<36}{"Output Shape":<18}{"Param #":>10}')
        print_fn('=' * 65)
        width = self.gene_size
        for lay in self.layers:
            width = lay.units if lay.units is not None else width
            print_fn(f'{lay.name + " (" + lay.kind + ")":<36}{str((None, width)):<18}{lay.count_params():>10}')
        print_fn('=' * 65)
        print_fn(f'Total params: {self.count_params():,}')


def build_cont_encoder(encoding_size, gene_size):
    """encoder_factory for ContinuousCellBiGan (reference src/bigan_cont.py:28-41)."""
    return HipEncoder(encoding_size, gene_size, ARCH_CONT)


def build_classify_encoder(encoding_size, gene_size):
    """encoder_factory for ClassifyCellBiGan (reference src/bigan_classify.py:28-40)."""
    return HipEncoder(encoding_size, gene_size, ARCH_CLS)
