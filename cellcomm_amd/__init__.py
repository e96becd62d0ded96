"""cellcomm_amd — MI355X-native encode + kNN hot path of mikemey/cellcomm.

Python call surface of the reference (``load_matrix``, ``CellTraining``,
``BasicBiGan.encoding_prediction``, encoder-factory injection) in front of
hand-written HIP kernels reached through the C ABI in include/cellcomm_hip.h.
There is no CPU fallback: anything that computes raises ``HipUnavailable`` when
libcellcomm_hip.so or a gfx950 device is missing.
"""
from ._lib import HipUnavailable, lib_path  # noqa: F401
from .matrix import CellMatrix, load_matrix, load_cells, store_cells, store_converted_cell_file  # noqa: F401
from .encoder import HipEncoder, build_cont_encoder, build_classify_encoder  # noqa: F401
from .generator import HipGenerator, build_cont_generator, build_classify_generator  # noqa: F401
from .bigan import BasicBiGan, ClassifyCellBiGan, ContinuousCellBiGan  # noqa: F401
from .training import CellTraining  # noqa: F401
from .knn import knn_graph  # noqa: F401
from .recorder import DbRecorder, MemoryDatabase  # noqa: F401
