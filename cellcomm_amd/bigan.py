"""BiGAN wrappers of the encode path: BasicBiGan, ClassifyCellBiGan, ContinuousCellBiGan.

The reference's seam on this path is constructor injection of component
factories (reference src/bigan_basic.py:11-21): ``BasicBiGan`` builds generator,
encoder and discriminator by calling the three factories with
``(encoding_size, gene_size)``, snapshots their Dense weights, and from then on
``encoding_prediction(cell_data)`` is nothing but ``_encoder.predict(cell_data)``
(:28-30, pinned by reference test/bigans_basic_test.py:25-31).  These classes keep
that contract with a HIP encoder as the default factory product.

Only inference of the encoder is on the MI355X hot path.  The discriminator is a
GAN-training component (out of scope) and the generator is built only
on request (``cellcomm_amd.generator`` has GPU factories for it): the default
factories return ``AbsentComponent`` placeholders which satisfy the constructor and
fail loudly when used; real components can still be injected.
"""
import numpy as np

from .encoder import build_classify_encoder, build_cont_encoder


class AbsentComponent:
    """Stands where a generator / discriminator would be."""

    def __init__(self, name, encoding_size, gene_size):
        self.name = name
        self.layers = []
        self.trainable = False
        self.encoding_size, self.gene_size = encoding_size, gene_size

    def summary(self):
        print(f'Model: "{self.name}"  [not built: GAN training is outside the MI355X hot path]')

    def predict(self, *args, **kwargs):
        raise NotImplementedError(f'{self.name} is not part of the encode + kNN hot path; inject a real '
                                  f'component through the factory arguments to use it')


def _absent(name):
    return lambda encoding_size, gene_size: AbsentComponent(name, encoding_size, gene_size)


def _dense_weight_snapshot(component):
    """Copies of the [kernel, bias] pairs of every two-weight layer — what the
    reference records per component at construction (src/bigan_basic.py:72-81)."""
    snap = []
    for layer in component.layers:
        ws = layer.get_weights()
        if len(ws) == 2:
            snap.extend(np.array(w, copy=True) for w in ws)
    return snap


def to_categorical(ixs, num_classes):
    ixs = np.asarray(ixs, dtype=np.int64)
    onehot = np.zeros((len(ixs), num_classes), dtype=np.float32)
    onehot[np.arange(len(ixs)), ixs] = 1.0
    return onehot


class BasicBiGan:
    def __init__(self, encoding_size, gene_size, generator_factory, encoder_factory, discriminator_factory):
        self.encoding_size = encoding_size
        self._generator = generator_factory(encoding_size, gene_size)
        self._encoder = encoder_factory(encoding_size, gene_size)
        self._discriminator = discriminator_factory(encoding_size, gene_size)
        self.all_components = self._generator, self._encoder, self._discriminator
        self._param_snapshot = [_dense_weight_snapshot(c) for c in self.all_components]

    def summary(self):
        for component in self.all_components:
            component.summary()

    def encoding_prediction(self, cell_data):
        return self._encoder.predict(cell_data)

    def random_uniform_vector(self, batch_size):
        # reference src/bigan_basic.py:35-36 (tf.random.uniform there)
        return np.random.uniform(0.0, 1.0, size=(batch_size, self.encoding_size)).astype(np.float32)

    def random_encoding_vector(self, batch_size):
        raise NotImplementedError

    def generate_cells(self, encoding_in, random_in=None):
        # reference src/bigan_basic.py:39-44; works only with an injected generator
        noise = self.random_uniform_vector(len(encoding_in)) if random_in is None else random_in
        return np.round(self._generator.predict((encoding_in, noise)))

    def trainings_step(self, sampled_batch):
        raise NotImplementedError('GAN training is outside the MI355X hot path (SURVEY.md §8a)')

    def params_changed(self):
        """[generator, encoder, discriminator] — True where a Dense weight differs from
        the last snapshot (the probe behind reference print_params_changes, :66-70)."""
        now = [_dense_weight_snapshot(c) for c in self.all_components]
        changed = [any(not np.array_equal(a, b) for a, b in zip(n, o)) for n, o in zip(now, self._param_snapshot)]
        self._param_snapshot = now
        return changed


    def print_params_changes(self, msg):
        # reference src/bigan_basic.py:66-70, same line format
        changed = self.params_changed()
        print(msg, 'G|E|D changed:', f'{changed[0]:1} | {changed[1]:1} | {changed[2]:1},')


def components_changed(p_now, p_orig):
    """True when any array of two weight snapshots differs (reference src/bigan_basic.py:84-89)."""
    for n, o in zip(p_now, p_orig):
        if not np.array_equal(n, o):
            return True
    return False


class ClassifyCellBiGan(BasicBiGan):
    """Softmax-encoder variant (reference src/bigan_classify.py:80-124)."""

    def __init__(self, encoding_size, gene_size,
                 generator_factory=_absent('cell_generator'),
                 encoder_factory=build_classify_encoder,
                 discriminator_factory=_absent('cell_discriminator')):
        super().__init__(encoding_size, gene_size, generator_factory, encoder_factory, discriminator_factory)

    def random_encoding_vector(self, batch_size):
        # one-hot rows from np.random.randint, as reference :117-119 (same draws under np.random.seed)
        return to_categorical(np.random.randint(0, self.encoding_size, batch_size), self.encoding_size)

    def trainings_encoding_prediction(self, cell_data):
        # argmax -> one-hot, reference :121-124
        return to_categorical(np.argmax(self.encoding_prediction(cell_data), -1), self.encoding_size)


class ContinuousCellBiGan(ClassifyCellBiGan):
    """Sigmoid-encoder variant used by the trainer (reference src/bigan_cont.py:44-56,
    src/cell_type_training.py:34)."""

    def __init__(self, encoding_size, gene_size, *, generator_factory=None):
        # generator_factory (additive to the reference signature): pass
        # cellcomm_amd.generator.build_cont_generator to make generate_cells run on the GPU; by
        # default the generator is not built (it holds 0.22·gene_size² weights).
        super().__init__(encoding_size, gene_size,
                         generator_factory=generator_factory or _absent('cell_generator'),
                         encoder_factory=build_cont_encoder)

    def random_encoding_vector(self, batch_size):
        return self.random_uniform_vector(batch_size)

    def trainings_encoding_prediction(self, cell_data):
        return self.encoding_prediction(cell_data)
