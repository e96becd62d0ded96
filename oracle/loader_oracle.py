"""Oracle: matrix loader and minibatch sampling.  TEST INFRASTRUCTURE ONLY.

Restates reference src/cell_type_training.py:9-17 (load_matrix: skip exactly 3
lines, whitespace triplets ``gene barcode p``, pandas pivot_table with
fill_value=0 — duplicates averaged, axes sorted) and :37-38
(DataFrame.sample(batch_size, random_state)).  Pinned by the reference's known
answers: test/cell_type_training_test.py:14-21,32-41.
"""
import pandas as pd


def load_matrix(matrix_file):
    df = pd.read_csv(matrix_file, header=None, skiprows=3, sep=r'\s+', names=['gene', 'barcode', 'p'])
    return df.pivot_table(index='barcode', columns='gene', values='p', fill_value=0)


def sample_cell_data(df, batch_size, random_seed=None):
    return df.sample(batch_size, random_state=random_seed)
