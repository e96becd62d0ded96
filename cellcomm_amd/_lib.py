"""ctypes binding of libcellcomm_hip.so (C ABI: include/cellcomm_hip.h).

This is the stub a reference maintainer would add behind the encoder factory
(reference src/bigan_basic.py:17-18); see INTEGRATION.md.  Loading is lazy so
that host-only logic (loader, schema builders) imports on a CPU-only machine,
but every compute entry point goes through ``lib()`` / ``ensure_init()`` which
raise ``HipUnavailable`` — there is no fallback path.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_INIT_DEVICE = None

PROF_CLASSES = ['gather', 'gemm_l1', 'gemm_l2', 'gemm_tail', 'softmax',
                'knn_pack', 'knn_select', 'knn_rerank', 'knn_fallback', 'spmm_l1', 'spmm_l2']


class HipUnavailable(RuntimeError):
    """The HIP library or a gfx950 device is missing; nothing was computed."""


def lib_path():
    return os.path.join(_HERE, 'libcellcomm_hip.so')


# name -> (restype, argtypes); must list every symbol declared in include/cellcomm_hip.h
SIGNATURES = {
    'cc_init': (c_int, [c_int]),
    'cc_shutdown': (c_int, []),
    'cc_version': (c_int, []),
    'cc_last_error': (c_char_p, []),
    'cc_device_count': (c_int, []),
    'cc_device_name': (c_int, [c_char_p, c_int]),
    'cc_synchronize': (c_int, []),
    'cc_encoder_create': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    'cc_encoder_free': (None, [c_void_p]),
    'cc_encoder_num_weights': (c_int, [c_void_p]),
    'cc_encoder_weight_shape': (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64)]),
    'cc_encoder_set_weights': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64]),
    'cc_encoder_set_batch': (c_int, [c_void_p, c_int64]),
    'cc_encoder_batch': (c_int64, [c_void_p]),
    'cc_encoder_fp32_fallback_passes': (c_int64, [c_void_p]),
    'cc_encoder_set_sparse_input': (c_int, [c_void_p, c_int]),
    'cc_encoder_set_low_latency': (c_int, [c_void_p, c_int]),
    'cc_generator_create': (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    'cc_generator_free': (None, [c_void_p]),
    'cc_generator_num_weights': (c_int, [c_void_p]),
    'cc_generator_weight_shape': (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64)]),
    'cc_generator_set_weights': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64]),
    'cc_generate_dense': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'cc_csr_upload': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, POINTER(c_void_p)]),
    'cc_csr_wrap_device': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, POINTER(c_void_p)]),
    'cc_csr_free': (None, [c_void_p]),
    'cc_gather_dense': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int]),
    'cc_encode': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'cc_encode_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'cc_encode_dense': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'cc_mtx_load': (c_int, [c_char_p, POINTER(c_void_p)]),
    'cc_mtx_shape': (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    'cc_mtx_export': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cc_mtx_free': (None, [c_void_p]),
    'cc_knn_l2': (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cc_knn_l2_dev': (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    'cc_knn_create': (c_int, [POINTER(c_void_p)]),
    'cc_knn_free': (None, [c_void_p]),
    'cc_knn_search_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    'cc_knn_set_option': (c_int, [c_void_p, c_int, c_int]),
    'cc_knn_stat': (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    'cc_knn_force_fallback': (c_int, [c_int]),
    'cc_knn_force_full_select': (c_int, [c_int]),
    'cc_knn_use_fp16_collect': (c_int, [c_int]),
    'cc_knn_last_fallback_count': (c_int64, []),
    'cc_knn_use_fp64_prefilter': (c_int, [c_int]),
    'cc_comm_unique_id': (c_int, [c_void_p]),
    'cc_comm_init': (c_int, [c_int, c_int, c_void_p, POINTER(c_void_p)]),
    'cc_comm_free': (c_int, [c_void_p]),
    'cc_allgather_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'cc_knn_l2_sharded': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cc_knn_last_full_scan_count': (c_int64, []),
    'cc_duplicate_representatives': (c_int, [c_void_p, c_int64, c_void_p]),
    'cc_profile_enable': (c_int, [c_int]),
    'cc_profile_reset': (c_int, []),
    'cc_profile_get': (c_int, [c_int, POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
}


def lib():
    """Load the shared library (once).  torch is imported first so that both
    share one HIP runtime instance (same SONAME), which lets torch tensors'
    device pointers be passed straight through the C ABI."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise HipUnavailable(f'{path} not found: build it with `python -c "import __graft_entry__ as g; '
                                 f'g.build()"` or `make -C cellcomm_amd/csrc`. There is no CPU fallback.')
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing only; the library works without it
            pass
        handle = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(rc):
    if rc != 0:
        msg = lib().cc_last_error().decode('utf-8', 'replace')
        if rc == 1:
            raise ValueError(msg)
        raise RuntimeError(f'libcellcomm_hip error {rc}: {msg}')


def ensure_init(device=None):
    """Initialise the library on `device` (default: LOCAL_RANK or 0)."""
    global _INIT_DEVICE
    if device is None:
        device = _INIT_DEVICE if _INIT_DEVICE is not None else int(os.environ.get('LOCAL_RANK', '0'))
    if _INIT_DEVICE == device:
        return lib()
    handle = lib()
    if handle.cc_device_count() == 0:
        raise HipUnavailable('no HIP device visible: cellcomm_amd computes only on a gfx950 (MI355X) GPU '
                             'and has no CPU fallback')
    rc = handle.cc_init(device)
    if rc != 0:
        raise HipUnavailable(handle.cc_last_error().decode('utf-8', 'replace'))
    _INIT_DEVICE = device
    return handle


def shutdown():
    """Release the library context (streams, pooled events, default kNN scratch).  Handles created by the
    caller (encoders, matrices, kNN workspaces) must have been freed first; afterwards `ensure_init` may
    bind another device."""
    global _INIT_DEVICE
    if _LIB is not None:
        check(_LIB.cc_shutdown())
    _INIT_DEVICE = None


def profile_enable(on=True):
    check(ensure_init().cc_profile_enable(1 if on else 0))


def profile_reset():
    check(ensure_init().cc_profile_reset())


def profile_snapshot():
    """{class: {'launches', 'ms', 'work'}} accumulated since the last reset."""
    handle = ensure_init()
    out = {}
    for i, name in enumerate(PROF_CLASSES):
        n, ms, work = c_int64(), c_double(), c_double()
        check(handle.cc_profile_get(i, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work)))
        out[name] = {'launches': n.value, 'ms': ms.value, 'work': work.value}
    return out
