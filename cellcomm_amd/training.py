"""CellTraining — the trainer shell around the hot path
(reference src/cell_type_training.py:29-50).

``data`` is a CellMatrix (CSR) rather than a dense DataFrame; the sampled
minibatch is a row view that the GPU gathers from CSR.  ``run`` keeps the
reference's loop shape (iterations x batches_per_iteration training steps, then
the interceptor with the summed losses) although the training step itself is
outside the MI355X hot path.
"""
from .bigan import ContinuousCellBiGan
from .matrix import CellMatrix


class CellTraining:
    def __init__(self, data, batch_size, encoding_size, batches_per_iteration=10):
        if not isinstance(data, CellMatrix):
            dense = data.to_numpy() if hasattr(data, 'to_numpy') else data
            data = CellMatrix.from_dense(dense, index=getattr(data, 'index', None),
                                         columns=getattr(data, 'columns', None))
        self.data = data
        self.batch_size = batch_size
        self.batches_per_iteration = batches_per_iteration
        self.network = ContinuousCellBiGan(encoding_size, gene_size=data.shape[1])

    def sample_cell_data(self, random_seed=None):
        return self.data.sample(self.batch_size, random_state=random_seed)

    def run(self, iterations, interceptor=None):
        for it in range(iterations):
            totals = [0, 0, 0]
            for _ in range(self.batches_per_iteration):
                losses = self.network.trainings_step(self.sample_cell_data())
                totals = [t + l for t, l in zip(totals, losses)]
            if interceptor:
                interceptor(it, tuple(totals))
