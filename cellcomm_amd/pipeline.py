"""encode -> (all-gather) -> kNN: the whole hot path, single GPU or cell-sharded.

Cells are independent through gather + encoder, so rows shard across the GPUs of
a node (one process per GPU); the only exchange is ONE all-gather of the
per-shard encodings (RCCL over xGMI through torch.distributed's "nccl"
backend) before each rank searches its own queries against all candidates
(SURVEY.md §8e).  No weight or gradient traffic, no all-reduce.
"""
import numpy as np


def shard_bounds(n_total, world_size, rank):
    """Contiguous row block of rank: ceil(n/world) rows each, tail ranks short/empty."""
    per = -(-n_total // world_size) if world_size > 0 else n_total
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


class HipBackend:
    """Compute backend of the product path: HIP kernels through the C ABI, device
    buffers as torch tensors.  (Tests inject a different backend to exercise the
    sharding logic on CPU with gloo; the product never does.)"""
    device = 'cuda'

    def __init__(self):
        import torch
        from . import _lib
        _lib.ensure_init()
        self.torch = torch
        # tensors handed to torch.distributed live on the GPU (RCCL); with a gloo group (CPU-only
        # rehearsal of the multi-rank path on one GPU) they are staged through host memory
        self.collective_device = 'cuda'
        if torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_backend() == 'gloo':
            self.collective_device = 'cpu'
        self._out_key, self._out = None, None
        self._side = None
        torch.cuda.set_device(_lib._INIT_DEVICE)

    def alloc(self, rows, width):
        return self.torch.empty((rows, width), dtype=self.torch.float32, device='cuda')

    def encode_into(self, encoder, matrix, out):
        n = matrix.shape[0]
        if n:
            encoder.predict_to_device(matrix, out.data_ptr(), 0, n)

    def copy_out(self, dev_rows, host_pinned):
        """Start the device-to-host copy of finished encodings on a side stream; returns the function that
        waits for it.  (`encode_into` has returned, so the rows are complete.)"""
        torch = self.torch
        if self._side is None:
            self._side = torch.cuda.Stream()
        with torch.cuda.stream(self._side):
            host_pinned.copy_(dev_rows, non_blocking=True)
        return self._side.synchronize

    def knn(self, enc_all, n_total, q_begin, n_q, k, include_self):
        from .knn import knn_graph_device
        # the all-gather ran on torch's current stream: wait for THAT stream only — a device-wide synchronize would also
        # wait for the copy of the encodings to host memory that `copy_out` started on its side stream, which is meant
        # to run underneath the search (round 3: 1.7 ms of the 209 ms step)
        self.torch.cuda.current_stream().synchronize()
        return knn_graph_device(enc_all.data_ptr(), n_total, enc_all.shape[1], k, q_begin, n_q, include_self,
                                out=self.result_buffers(n_q, k))

    def result_buffers(self, n_q, k):
        """Pinned host arrays for the neighbour ids and distances of a search, reused across calls."""
        key = (n_q, k)
        if self._out_key != key:
            self._out = (self.torch.empty(key, dtype=self.torch.int32, pin_memory=True),
                         self.torch.empty(key, dtype=self.torch.float32, pin_memory=True))
            self._out_key = key
        return self._out[0].numpy(), self._out[1].numpy()


def encode_and_knn_native(encoder, local_matrix, k, comm, n_total, include_self=False, backend=None, host_encodings=None,
                          timings=None, fused=False):
    """The same path with the exchange done by the library itself (`comm.NativeComm`: RCCL through the C ABI,
    no torch.distributed in the data path).  `local_matrix` must be block `shard_bounds(n_total, comm.world_size,
    comm.rank)` of the global matrix.  Returns the same dict as `encode_and_knn` plus `enc_local` (this rank's block,
    padded to ceil(n_total / world) rows).

    `fused=False`: `cc_allgather_f32` into a buffer of the caller, then the search (`cc_knn_l2_dev`).
    `fused=True`: ONE library call does both (`cc_knn_l2_sharded`, csrc/comm.cpp — what a non-Python host calls); the
    gathered encodings then stay inside the communicator and `encodings` is None.
    `host_encodings` / `timings`: as in `encode_and_knn` (`allgather_ms` is part of `knn_ms` when fused)."""
    import time
    clock = time.perf_counter

    def tick(key, t_start):
        if timings is not None:
            timings[key] = timings.get(key, 0.0) + (clock() - t_start) * 1e3
    backend = backend or HipBackend()
    lo, hi = shard_bounds(n_total, comm.world_size, comm.rank)
    n_local = local_matrix.shape[0]
    if n_local != hi - lo:
        raise ValueError(f'rank {comm.rank} of {comm.world_size} must hold rows [{lo}, {hi}) of {n_total}, got {n_local} rows')
    width = encoder.encoding_size
    per = -(-n_total // comm.world_size)
    enc_local = backend.alloc(per, width)
    if n_local < per:
        enc_local[n_local:].zero_()
        backend.torch.cuda.synchronize()
    t0 = clock()
    backend.encode_into(encoder, local_matrix, enc_local)
    tick('encode_ms', t0)
    done = backend.copy_out(enc_local[:n_local], host_encodings) if host_encodings is not None else None
    if fused:
        t0 = clock()
        idx, dst = comm.knn_sharded(enc_local.data_ptr(), n_local, n_total, width, k, include_self,
                                    out=backend.result_buffers(n_local, k))
        tick('knn_ms', t0)
        enc_all = None
    else:
        gathered = backend.alloc(comm.world_size * per, width)
        backend.torch.cuda.current_stream().synchronize()
        t0 = clock()
        comm.all_gather(enc_local.data_ptr(), per * width, gathered.data_ptr())     # the one collective
        tick('allgather_ms', t0)
        enc_all = gathered[:n_total]
        t0 = clock()
        idx, dst = backend.knn(enc_all, n_total, lo, n_local, k, include_self)
        tick('knn_ms', t0)
    if done is not None:
        done()
    return dict(encodings=enc_all, idx=idx, dist=dst, q_begin=lo, n_total=n_total, enc_local=enc_local)


def encode_and_knn(encoder, local_matrix, k, include_self=False, backend=None, group=None, host_encodings=None,
                   n_total=None, timings=None):
    """Run the hot path on this rank's block of cells.

    `local_matrix` holds the rows of this rank (block `shard_bounds(n_total, world,
    rank)` of the global matrix; the whole matrix when not distributed).
    Returns dict(encodings=[n_total x E] tensor on the backend device (all ranks),
    idx=[n_local x k] int32 global neighbour ids, dist=[n_local x k] float32 (with
    HipBackend these are views of pinned buffers that the next call overwrites),
    q_begin=first global row of this rank, n_total=...).
    `host_encodings` (optional, a pinned [n_local x E] float32 torch tensor, HipBackend only) receives this
    rank's encodings; the device-to-host copy runs on a side stream underneath the kNN search.
    `n_total` (optional): total number of cells when every rank holds block `shard_bounds(n_total, world, rank)`;
    saves the small collective that otherwise exchanges the block sizes each call.
    `timings` (optional dict): host-clock milliseconds of this rank are ADDED to its keys `encode_ms`, `allgather_ms`
    (from "my encodings are complete" to "everybody's are here": the collective plus the wait for the slowest rank)
    and `knn_ms` — how a bad scaling curve is read (bench.py `stages_by_rank`).
    """
    import time
    clock = time.perf_counter

    def tick(key, t_start):
        if timings is not None:
            timings[key] = timings.get(key, 0.0) + (clock() - t_start) * 1e3
    backend = backend or HipBackend()
    torch = backend.torch
    dist = torch.distributed
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    n_local = local_matrix.shape[0]
    width = encoder.encoding_size

    if not distributed:
        enc_all = backend.alloc(n_local, width)
        t0 = clock()
        backend.encode_into(encoder, local_matrix, enc_all)
        tick('encode_ms', t0)
        done = backend.copy_out(enc_all, host_encodings) if host_encodings is not None else None
        t0 = clock()
        idx, dst = backend.knn(enc_all, n_local, 0, n_local, k, include_self)
        tick('knn_ms', t0)
        if done is not None:
            done()
        return dict(encodings=enc_all, idx=idx, dist=dst, q_begin=0, n_total=n_local)

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cdev = getattr(backend, 'collective_device', backend.device)
    if n_total is not None:
        counts = np.array([shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)],
                          dtype=np.int64)
        if int(counts[rank]) != n_local:
            raise ValueError(f'rank {rank} of {world} must hold {int(counts[rank])} of {n_total} rows, got {n_local}')
    else:
        counts = torch.zeros(world, dtype=torch.int64, device=cdev)
        mine = torch.tensor([n_local], dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(counts, mine, group=group)
        counts = counts.cpu().numpy()
    n_total = int(counts.sum())
    per = int(counts.max())
    offsets = np.concatenate([[0], np.cumsum(counts)])
    q_begin = int(offsets[rank])

    enc_local = backend.alloc(per, width)
    if n_local < per:
        enc_local[n_local:].zero_()
    t0 = clock()
    backend.encode_into(encoder, local_matrix, enc_local)
    tick('encode_ms', t0)
    done = backend.copy_out(enc_local[:n_local], host_encodings) if host_encodings is not None else None
    gathered = backend.alloc(world * per, width)
    if backend.device == 'cuda':
        # the encoder call returned synchronised (library stream); what may still be pending on torch's stream is the
        # zero fill of a short block's padding rows.  Not a device-wide synchronize: the copy of the encodings to the
        # host (side stream) is meant to run underneath the collective and the search
        torch.cuda.current_stream().synchronize()
    t0 = clock()
    if cdev == backend.device:
        dist.all_gather_into_tensor(gathered, enc_local, group=group)  # the one collective
    else:
        staged = torch.empty((world * per, width), dtype=torch.float32)
        dist.all_gather_into_tensor(staged, enc_local.cpu(), group=group)
        gathered.copy_(staged)
    if timings is not None:
        if backend.device == 'cuda':
            torch.cuda.current_stream().synchronize()   # (the search below starts with the same synchronisation)
        tick('allgather_ms', t0)
    if all(int(c) == per for c in counts[:-1]):
        enc_all = gathered[:n_total]            # blocks are already back to back
    else:
        enc_all = torch.cat([gathered[r * per:r * per + int(counts[r])] for r in range(world)]).contiguous()
    t0 = clock()
    idx, dst = backend.knn(enc_all, n_total, q_begin, n_local, k, include_self)
    tick('knn_ms', t0)
    if done is not None:
        done()
    return dict(encodings=enc_all, idx=idx, dist=dst, q_begin=q_begin, n_total=n_total)
