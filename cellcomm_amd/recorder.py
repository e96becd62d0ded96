"""The run recorder of the reference (src/intercepts/db_recorder.py:28-110) without a Mongo server.

`DbRecorder(enc_run_id, sources, m_db)` keeps the reference's names, call order and assertion
messages: `setup()` = `store_encoding_run()` (the run-level `encs` document
`{_id, date, defit, showits, srcs}` that cellan fetches first, cellan/backend/apiRouter.js:8-15) +
`load_barcodes()` (fills `cells` / `genes` for the source on first use, import_barcodes.py:64-81);
`create_interceptor(trainer)` returns the `(it, losses)` closure that stores one `encits` document per
iteration and moves `defit` / `showits` of the run document along (db_recorder.py:82-108).

`m_db` is any object that maps a collection name to a collection with the five pymongo calls the
reference uses (`find_one`, `insert_one`, `insert_many`, `update_one` with `$set`, `count_documents`,
`find` with a projection): a `pymongo` database works unchanged, `MemoryDatabase` (default) keeps the
documents in process and can dump them as JSON for `mongoimport`.
"""
import copy
import json
import os
from datetime import datetime

import numpy as np

from . import schema

ENCODINGS_COLLECTION = 'encs'
ITERATIONS_COLLECTION = 'encits'
CELLS_COLLECTION = 'cells'
GENES_COLLECTION = 'genes'


def get_file_name(full_path):
    return full_path.split('/')[-1]


def check_files(sources):
    missing = [src for src in sources if not os.path.exists(src)]
    if missing:
        raise AssertionError(f'File not found: {missing[0]}')


def _matches(doc, query):
    return all(doc.get(k) == v for k, v in query.items())


def _project(doc, projection):
    if not projection:
        return copy.deepcopy(doc)
    keep = [k for k, v in projection.items() if v]
    if keep:
        return {k: copy.deepcopy(doc[k]) for k in keep if k in doc}
    drop = {k for k, v in projection.items() if not v}
    return {k: copy.deepcopy(v) for k, v in doc.items() if k not in drop}


class MemoryCollection:
    """The subset of `pymongo.collection.Collection` the recorder touches, over a Python list
    (insertion order = natural order, equality matches on top-level fields only)."""

    def __init__(self):
        self.docs = []

    def find_one(self, query=None, projection=None):
        for doc in self.docs:
            if _matches(doc, query or {}):
                return _project(doc, projection)
        return None

    def find(self, query=None, projection=None):
        return [_project(doc, projection) for doc in self.docs if _matches(doc, query or {})]

    def count_documents(self, query):
        return sum(1 for doc in self.docs if _matches(doc, query))

    def insert_one(self, doc):
        if '_id' in doc and any(d.get('_id') == doc['_id'] for d in self.docs):
            raise KeyError(f'duplicate _id {doc["_id"]!r}')
        self.docs.append(copy.deepcopy(doc))

    def insert_many(self, docs):
        for doc in docs:
            self.insert_one(doc)

    def update_one(self, query, update):
        for doc in self.docs:
            if _matches(doc, query):
                doc.update(copy.deepcopy(update.get('$set', {})))
                return


class MemoryDatabase:
    def __init__(self):
        self.collections = {}

    def __getitem__(self, name):
        return self.collections.setdefault(name, MemoryCollection())

    def dump(self, directory):
        """One `<collection>.json` (JSON lines, dates as ISO strings) per collection — `mongoimport`'s input."""
        os.makedirs(directory, exist_ok=True)
        for name, coll in self.collections.items():
            with open(os.path.join(directory, f'{name}.json'), 'w') as f:
                for doc in coll.docs:
                    f.write(json.dumps(doc, default=lambda o: o.isoformat() if isinstance(o, datetime) else str(o)) + '\n')


def _require(condition, message):
    """The reference signals misuse with `assert` (AssertionError + message); same type and text here, also under -O."""
    if not condition:
        raise AssertionError(message)


class _IterationWriter:
    """The `(it, losses)` interceptor of one recorder: encodes the trainer's whole data set, stores the iteration's
    `encits` document and moves the run document's `defit` / `showits` along."""

    def __init__(self, owner, trainer, dims, knn_k):
        self.owner, self.trainer, self.dims, self.knn_k = owner, trainer, dims, knn_k
        self.shown = []

    def __call__(self, it, _losses):
        rec = self.owner
        _require(it not in rec._seen_iterations, f'duplicate iteration {it}')
        rec._seen_iterations.append(it)
        encodings = self.trainer.network.encoding_prediction(self.trainer.data)
        graph = None
        if self.knn_k is not None:
            from .knn import knn_graph
            graph = knn_graph(np.ascontiguousarray(encodings, dtype=np.float32), self.knn_k)
        if self.dims is None:
            document = schema.encits_document(rec.enc_run_id, it, encodings, rec.barcodes, knn=graph)
        else:
            document = schema.encits_from_wide(rec.enc_run_id, it, encodings, rec.barcodes, dims=self.dims, knn=graph)
        document['cids'] = rec.cell_ids
        rec._collection(ITERATIONS_COLLECTION).insert_one(document)
        self.shown.append(it)
        rec._collection(ENCODINGS_COLLECTION).update_one({'_id': rec.enc_run_id},
                                                         {'$set': {'defit': it, 'showits': list(self.shown)}})


class DbRecorder:
    """`DbRecorder(enc_run_id, sources, m_db)`; `sources` = {'matrix', 'barcodes', 'genes'} file paths."""

    def __init__(self, enc_run_id, sources, m_db=None):
        self.enc_run_id = enc_run_id
        self.matrix_file, self.barcodes_file, self.genes_file = (sources[k] for k in ('matrix', 'barcodes', 'genes'))
        check_files([self.matrix_file, self.barcodes_file, self.genes_file])
        self.mongo_db = MemoryDatabase() if m_db is None else m_db
        self.source_id = self.barcodes = self.cell_ids = None
        self._seen_iterations = []

    def _collection(self, name):
        return self.mongo_db[name]

    def setup(self):
        self.store_encoding_run()
        self.load_barcodes()

    def store_encoding_run(self, now=None):
        runs = self._collection(ENCODINGS_COLLECTION)
        _require(runs.find_one({'_id': self.enc_run_id}) is None, f'Encoding run id already exists: {self.enc_run_id}')
        self.source_id = get_file_name(self.barcodes_file)
        paths = {'matrix': self.matrix_file, 'barcodes': self.barcodes_file, 'genes': self.genes_file}
        runs.insert_one(schema.encs_run_document(self.enc_run_id, paths, now))

    def load_barcodes(self):
        _require(self.source_id, 'Cannot load barcodes without encoding!')
        cells = self._collection(CELLS_COLLECTION)
        by_source = {'sid': self.source_id}
        if cells.count_documents(by_source) == 0:      # first run on this source: import its cells / genes documents
            cell_docs, gene_docs = schema.cells_and_genes_documents(self.source_id, self.matrix_file, self.barcodes_file,
                                                                    self.genes_file)
            cells.insert_many(cell_docs)
            self._collection(GENES_COLLECTION).insert_many(gene_docs)
        self.barcodes = [doc['n'] for doc in cells.find(by_source, {'_id': 0, 'n': 1})]
        self.cell_ids = [schema.cell_id_from(pos) for pos in range(len(self.barcodes))]

    def create_interceptor(self, trainer, dims=None, knn_k=None):
        """`dims=None` is the reference: the encodings must be N x 3.  `dims=(a, b, c)` exports those three
        latent dimensions of wider encodings (E = 32/64; `schema.encits_from_wide`).  `knn_k` attaches the exact
        kNN graph of the FULL-width encodings as the additive `nn` / `nd` fields."""
        _require(self.barcodes, 'Cannot store iterations without barcodes!')
        return _IterationWriter(self, trainer, dims, knn_k)
