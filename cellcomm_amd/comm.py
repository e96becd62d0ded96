"""The sharded path's exchange through the library's own C ABI (RCCL bound at run time).

`pipeline.encode_and_knn` uses torch.distributed's "nccl" backend (= RCCL) by default because that is how
the ranks of a node are launched here; `NativeComm` is the same all-gather without PyTorch in the data path
— what a non-Python host would call (`cc_comm_init`, `cc_knn_l2_sharded`; include/cellcomm_hip.h).
"""
import ctypes

import numpy as np

from . import _lib

UNIQUE_ID_BYTES = 128


def make_unique_id():
    """128-byte RCCL id; rank 0 calls this and the host program distributes the bytes to every rank."""
    _lib.ensure_init()
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    _lib.check(_lib.lib().cc_comm_unique_id(buf))
    return buf.raw


class NativeComm:
    """One rank of an RCCL communicator (one process per GPU)."""

    def __init__(self, world_size, rank, unique_id):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError(f'unique id must be {UNIQUE_ID_BYTES} bytes, got {len(unique_id)}')
        _lib.ensure_init()
        self.world_size, self.rank = int(world_size), int(rank)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().cc_comm_init(self.world_size, self.rank, ctypes.c_char_p(bytes(unique_id)),
                                           ctypes.byref(self._h)))

    @classmethod
    def from_torch_distributed(cls, group=None):
        """Bootstrap from an initialised torch.distributed group: rank 0's id is broadcast as an object."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [make_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(world, rank, box[0])

    def all_gather(self, dev_send_ptr, count, dev_recv_ptr):
        """recv[world x count] <- every rank's send[count] (float32, device pointers)."""
        _lib.check(_lib.lib().cc_allgather_f32(self._h, ctypes.c_void_p(dev_send_ptr), int(count),
                                               ctypes.c_void_p(dev_recv_ptr)))

    def knn_sharded(self, dev_local_ptr, n_local, n_total, d, k, include_self=False, out=None):
        """All-gather the per-rank encodings and search this rank's rows against all of them."""
        if out is None:
            out = (np.empty((n_local, k), dtype=np.int32), np.empty((n_local, k), dtype=np.float32))
        idx, dist = out
        rc = _lib.lib().cc_knn_l2_sharded(self._h, ctypes.c_void_p(dev_local_ptr), int(n_local), int(n_total), int(d),
                                          int(k), 1 if include_self else 0, idx.ctypes.data, dist.ctypes.data)
        _lib.check(rc)
        return idx, dist

    def close(self):
        if self._h:
            _lib.lib().cc_comm_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
