#!/usr/bin/env python3
"""Headline benchmark: cells/sec of encode + kNN (BASELINE.json metric).

Workload (config 3 of BASELINE.json): synthetic 100k cells x 30k genes (5 % dense
integer counts), `cont` BiGAN encoder (30000 -> 3000 -> [3000+30000] -> 1500 ->
BN -> 150 -> 150 -> 64, random Glorot weights), exact k=15 kNN on the encodings.
A step = one pass of the whole hot path: device-resident CSR -> gather -> encoder
-> (all-gather) -> kNN -> encodings + neighbour ids on the host.

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N workers)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With N > 1 the 100k cells are split into N contiguous blocks (strong scaling), one
process per GPU, one RCCL all-gather of encodings per step.  Called without a
torch.distributed environment, `--gpus N` launches `python -m torch.distributed.run`
as a CHILD process before this process touches torch or the GPU, relays rank 0's
JSON line and exits with the child's status.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix (= fp32 vector peak)
PEAK_F16_MFMA_TFLOPS = 2500.0    # dense fp16/bf16 matrix
PEAK_HBM_GBS = 8000.0
PEAK_LDS_TBS = 157.3             # ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz (the guide's LDS table; ~150 TB/s streaming)
PCIE_GBS = 63.0                  # host link (spec), bounds the dense-host side numbers
TRAFFIC_PROFILES = ('r04_pmc_hbm_traffic.json', 'r03_pmc_hbm_traffic.json', 'r02_pmc_hbm_traffic.json')   # newest tracked PMC summary present   # tracked PMC summary that `roofline.traffic` is replayed from


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, 'BASELINE.json')))['metric']
    except Exception:
        return 'cells/sec encode+kNN, 100k cells × 30k genes, 1/2/4/8 MI355X'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--cells', type=int, default=100000)
    ap.add_argument('--genes', type=int, default=30000)
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--k', type=int, default=15)
    ap.add_argument('--arch', default='cont')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'f16x2'],
                    help='arithmetic of encoder layers 1-2: exact fp32 MFMA (default), fp16 operands, fp16 hi+lo split')
    ap.add_argument('--batch-rows', type=int, default=0, help='rows per pass through the layer stack (0 = library default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true',
                    help='skip the extra measurements reported under "alt" / "alt_f16" / "alt_sparse" / "dense_host"')
    ap.add_argument('--alt-multi', action='store_true', help='run the "alt*" legs with --gpus N > 1 as well (default: N = 1 only)')
    ap.add_argument('--no-dense-host', action='store_true',
                    help='skip the dense-host side numbers (cc_encode_dense throughput, batch-32/128 latency); tools/profile_round.sh '
                         'passes it so that the per-kernel averages of the profiled run are those of the timed passes')
    ap.add_argument('--dense-host', action='store_true', help='(default now; kept for old command lines)')
    ap.add_argument('--sparse', action='store_true', help='make the sparse first layer (fp32 over the non-zeros) the measured path')
    ap.add_argument('--no-profile', action='store_true', help='diagnostic: no HIP events in the timed region')
    ap.add_argument('--native-comm', action='store_true',
                    help='the exchange through the library itself (cc_knn_l2_sharded: RCCL bound by csrc/comm.cpp, no '
                         'torch.distributed in the data path) instead of torch.distributed all_gather_into_tensor')
    ap.add_argument('--no-parity-check', action='store_true',
                    help='skip the parity evidence taken after the timed region (gathered rows re-encoded, sampled queries vs the oracle)')
    ap.add_argument('--cpu-sample-cells', type=int, default=8192)
    ap.add_argument('--cpu-sample-knn', type=int, default=40000)
    return ap.parse_args()


def workload_name(args, world):
    """Which BASELINE.json config the arguments are (configs[i] as listed there), or 'custom'."""
    shape = (args.cells, args.genes, args.latent, args.k, args.arch)
    if shape == (100000, 30000, 64, 15, 'cont'):
        return 'BASELINE configs[2]'
    if shape[:3] == (10000, 20000, 32) and args.arch == 'cont':
        return 'BASELINE configs[1] (its encode; this bench also builds the kNN graph)'
    if shape == (1000000, 30000, 64, 15, 'cont'):
        return 'BASELINE configs[3]' + ('' if world == 8 else f' on {world} GPU(s) instead of 8')
    if shape == (125000, 30000, 64, 15, 'cont') and world == 1:
        return "custom: one rank's encode share of BASELINE configs[3] (125k of 1M cells; kNN among these rows only)"
    if shape == (500000, 33000, 64, 50, 'cont'):
        return 'BASELINE configs[4]' + ('' if world == 8 and args.dtype == 'f16' else f' with dtype {args.dtype} on {world} GPU(s)')
    if shape == (62500, 33000, 64, 50, 'cont') and world == 1:
        return "custom: one rank's encode share of BASELINE configs[4] (62.5k of 500k cells; kNN among these rows only)"
    return 'custom'


def cpu_baseline(args, enc, local, enc_host, alt_samples=None):
    """The oracle (a CPU port of the path; the reference's TF code cannot run here or on the
    GPU box) timed on this host's cores on a bounded sample of the same workload."""
    import ctypes
    from cellcomm_amd import _lib
    from oracle import encoder_oracle as eo
    from oracle import knn_oracle as ko
    ns = min(args.cpu_sample_cells, local.shape[0])
    ld = (args.genes + 31) // 32 * 32
    dense = np.empty((ns, ld), dtype=np.float32)
    _lib.check(_lib.lib().cc_gather_dense(local.device_handle(), None, 0, ns, ld, dense.ctypes.data, 0))
    dense = np.ascontiguousarray(dense[:, :args.genes])
    ws = enc.get_weights()
    t0 = time.perf_counter()
    out = eo.forward(args.arch, ws, dense, np.float32)              # one batch, all cores (BLAS sgemm)
    t_enc = time.perf_counter() - t0
    nb = min(256, ns)
    t0 = time.perf_counter()
    eo.predict_like_keras(args.arch, ws, dense[:nb])                # batches of 32, as keras predict
    t_b32 = (time.perf_counter() - t0) / nb
    nk = min(args.cpu_sample_knn, enc_host.shape[0])
    t0 = time.perf_counter()
    ko.knn(enc_host[:nk], args.k)
    t_knn = time.perf_counter() - t0
    per_pair = t_knn / (nk * nk)
    per_cell = t_enc / ns + per_pair * args.cells
    rel = float(np.max(np.abs(out - enc_host[:ns]) / np.abs(out))) if local.row_ids() is None else None
    # the other measured modes against the same CPU sample (the rows each timed leg left in host memory)
    alt_rel = {}
    for name, sample in (alt_samples or {}).items():
        m = min(ns, len(sample))
        alt_rel[name] = float(np.max(np.abs(out[:m] - sample[:m]) / np.abs(out[:m])))
        if name == 'alt_f16':   # ... and against the float64 evaluation of ITS arithmetic: operands rounded to fp16 once
            mf = min(128, m)
            ref16 = eo.forward_f16(args.arch, ws, dense[:mf])
            alt_rel['alt_f16_vs_f16_operand_oracle'] = float(np.max(np.abs(ref16 - sample[:mf]) / np.abs(ref16)))
    blas_threads = None
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max((p.get('num_threads', 0) for p in threadpool_info() if p.get('user_api') == 'blas'), default=None)
    except Exception:
        pass
    omp_threads = int(os.environ.get('OMP_NUM_THREADS', 0)) or os.cpu_count()
    return {
        'value': 1.0 / per_cell, 'unit': 'cells/s', 'cores': max(blas_threads or 0, omp_threads), 'kind': 'port',
        'blas_threads': blas_threads, 'openmp_threads': omp_threads, 'host_cpus': os.cpu_count(),
        'sample': f'oracle fp32 NumPy/BLAS forward on {ns} cells x {args.genes} genes in one batch '
                  f'({t_enc:.2f} s) + oracle/knn_oracle.c (OpenMP) on {nk} x {nk} encodings ({t_knn:.2f} s), '
                  f'kNN cost scaled per candidate pair to {args.cells} candidates',
        'encode_cells_per_s': ns / t_enc,
        'encode_cells_per_s_batch32_like_keras_predict': 1.0 / t_b32,
        'knn_pairs_per_s': 1.0 / per_pair,
        'gpu_vs_cpu_sample_max_rel': rel,
        'alt_max_rel': alt_rel,
    }


def parity_check(args, world, rank, enc, res, native, backend):
    """Parity evidence of THIS run, taken after the timed region (never inside it), on every rank:
    (1) the rank regenerates ONE other rank's block of cells (same generator, that rank's seed), encodes its first 256
        rows itself and requires them bit-identical to those rows of the GATHERED encodings — the all-gather delivered
        the other rank's bits to the right place (a row's encoding does not depend on the call it sits in);
    (2) 32 of its own queries against oracle/knn_oracle.c on the gathered encodings: neighbour ids and distances
        bit-exact, global numbering.
    The oracle is the checker here, exactly as in tests/; nothing it computes is timed or reported as a result."""
    import torch
    from cellcomm_amd import pipeline, synth
    from oracle import knn_oracle as ko
    out = {'rank': rank, 'gathered_rows_bit_identical': False, 'knn_queries_bit_exact': 0, 'knn_queries_checked': 0}
    lo, hi = pipeline.shard_bounds(args.cells, world, rank)
    enc_all = res['encodings']
    if enc_all is None:
        # the fused exchange keeps the gathered rows inside the communicator: gather once more (cc_allgather_f32)
        per = -(-args.cells // world)
        gathered = torch.empty((world * per, args.latent), dtype=torch.float32, device='cuda')
        torch.cuda.synchronize()
        native.all_gather(res['enc_local'].data_ptr(), per * args.latent, gathered.data_ptr())
        enc_all = gathered[:args.cells]
    other = (rank + 1) % world
    lo2, hi2 = pipeline.shard_bounds(args.cells, world, other)
    rows = min(256, hi2 - lo2)
    if rows > 0:
        blk, _keep = synth.synth_csr_device(hi2 - lo2, args.genes, density=0.05, seed=1234 + other)
        mine = torch.empty((rows, args.latent), dtype=torch.float32, device='cuda')
        enc.predict_to_device(blk, mine.data_ptr(), 0, rows)
        torch.cuda.synchronize()
        out['gathered_rows_bit_identical'] = bool(torch.equal(mine, enc_all[lo2:lo2 + rows]))
        out['rows_checked'], out['block_of_rank'] = rows, other
        del blk, _keep
    else:
        out['gathered_rows_bit_identical'] = True
    host_all = enc_all.cpu().numpy()
    n_local = hi - lo
    picks = np.random.default_rng(77 + rank).choice(n_local, min(32, n_local), replace=False) if n_local else []
    good = 0
    for q in picks:
        wi, wd = ko.knn(host_all, args.k, q_begin=lo + int(q), n_q=1)
        good += int(np.array_equal(res['idx'][q], wi[0]) and np.array_equal(res['dist'][q], wd[0]))
    out['knn_queries_bit_exact'], out['knn_queries_checked'] = good, len(picks)
    return out


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N workers as a child process group.
    Nothing in this process has imported torch or touched HIP yet, and nothing will."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY: inherited when the caller set it; otherwise 0 = dmabuf IPC.  The host driver of this
    # pool supports only dmabuf handles: with the legacy mode RCCL's intra-node P2P set-up (hipIpcGetMemHandle) and any
    # sharing of device tensors between processes fail with "invalid argument" (DESIGN.md §5).  Needed by the real
    # RCCL path, not only by the one-device rehearsal.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                     # rank 0 prints exactly one JSON line; anything else is chatter
        text = out.strip()
        if text.startswith('{') and '"metric"' in text:
            line = text
        elif text:
            print(text, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(f'bench.py --gpus {args.gpus}: worker group exited with status {rc}')
    if line is None:
        raise SystemExit(f'bench.py --gpus {args.gpus}: workers finished without a result line')
    print(line, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('CC_BENCH_ONE_DEVICE'):   # rehearsal: several ranks share GPU 0 (1-GPU box), gloo collective
        local_rank = 0
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # one process per GPU: fewer devices than ranks cannot work (RCCL refuses two ranks on one device) and must not
    # hang in a collective — every rank sees the same count and leaves before any rendezvous
    ndev = torch.cuda.device_count()
    if world > ndev and not os.environ.get('CC_BENCH_ONE_DEVICE'):
        raise SystemExit(f'bench.py rank {rank}: --gpus {world} needs {world} visible GPUs, this host shows {ndev} '
                         f'(one-GPU rehearsal of the multi-rank path: CC_BENCH_ONE_DEVICE=1, gloo collective)')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ.get('CC_BENCH_ONE_DEVICE'):
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    import cellcomm_amd as cc
    from cellcomm_amd import _lib, pipeline, synth
    _lib.ensure_init(local_rank)

    lo, hi = pipeline.shard_bounds(args.cells, world, rank)
    # each rank synthesises its own block of rows (same distribution, rank-specific seed); weights are identical
    local, _keep = synth.synth_csr_device(hi - lo, args.genes, density=0.05, seed=1234 + rank)
    enc = cc.HipEncoder(args.latent, args.genes, args.arch, seed=4321,
                        batch_rows=args.batch_rows or None, dtype=args.dtype, sparse=args.sparse)
    backend = pipeline.HipBackend()
    host_enc_t = torch.empty((hi - lo, args.latent), dtype=torch.float32, pin_memory=True)
    host_enc = host_enc_t.numpy()

    host_ms = {}   # host-clock milliseconds of this rank per stage, summed over the timed steps (pipeline `timings`)

    native = None
    if args.native_comm:
        from cellcomm_amd import comm as cccomm
        if world > 1 and os.environ.get('CC_BENCH_ONE_DEVICE'):
            raise SystemExit('bench.py: --native-comm needs one GPU per rank (RCCL refuses two ranks on one device); '
                             'the CC_BENCH_ONE_DEVICE rehearsal covers the torch.distributed exchange only')
        native = cccomm.NativeComm.from_torch_distributed() if world > 1 else cccomm.NativeComm(1, 0, cccomm.make_unique_id())

    def step():
        # encodings reach pinned host memory on a side stream while the kNN search runs
        if native is not None:
            return pipeline.encode_and_knn_native(enc, local, args.k, native, args.cells, backend=backend,
                                                  host_encodings=host_enc_t, timings=host_ms, fused=True)
        return pipeline.encode_and_knn(enc, local, args.k, backend=backend, host_encodings=host_enc_t,
                                       n_total=args.cells if world > 1 else None, timings=host_ms)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.profile_enable(not args.no_profile)
    _lib.profile_reset()
    fence()
    host_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_snapshot()
    _lib.profile_enable(False)
    main_host_ms = {k: v / args.steps for k, v in host_ms.items()}
    main_sample = host_enc.copy()   # the alt legs below reuse the pinned buffer
    # parity evidence of this very run (outside the timed region; before the alt legs overwrite the pinned results)
    parity = None
    if not args.no_parity_check:
        try:
            mine_parity = parity_check(args, world, rank, enc, res, native, backend)
        except Exception as err:    # a failing check must not leave the other ranks in a collective
            mine_parity = {'rank': rank, 'gathered_rows_bit_identical': False, 'knn_queries_bit_exact': 0,
                           'knn_queries_checked': -1, 'error': f'{type(err).__name__}: {err}'}
        everyone_parity = [mine_parity]
        if world > 1:
            everyone_parity = [None] * world
            dist.all_gather_object(everyone_parity, mine_parity)
        parity = {'gathered_rows_bit_identical': all(p['gathered_rows_bit_identical'] for p in everyone_parity),
                  'knn_queries_bit_exact': sum(p['knn_queries_bit_exact'] for p in everyone_parity),
                  'knn_queries_checked': sum(max(p['knn_queries_checked'], 0) for p in everyone_parity),
                  'ranks_checked': len(everyone_parity),
                  'exchange': 'cc_knn_l2_sharded (RCCL via csrc/comm.cpp)' if native is not None else
                              ('torch.distributed all_gather_into_tensor (%s)' % dist.get_backend() if world > 1 else 'none (one rank)'),
                  'rows_checked_per_rank': 256, 'queries_per_rank': 32,
                  'errors': [p['error'] for p in everyone_parity if 'error' in p],
                  'note': 'taken after the timed region: every rank re-encodes the first rows of ANOTHER rank\'s block and '
                          'compares them with the gathered encodings bit for bit, and checks 32 of its own queries against '
                          'oracle/knn_oracle.c on the gathered encodings (ids and distances bit-exact)'}
        parity['ok'] = bool(parity['gathered_rows_bit_identical'] and not parity['errors']
                            and parity['knn_queries_bit_exact'] == parity['knn_queries_checked'])
    stages_by_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=backend.collective_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per stage: slowest and fastest rank (kernel classes from the library's HIP events + the host-clock stages)
        mine = {k: v['ms'] / args.steps for k, v in prof.items() if v['launches']}
        mine.update({'host_' + k: v for k, v in main_host_ms.items()})
        mine['host_step_ms'] = sum(main_host_ms.values())
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        keys = sorted(set().union(*[set(d) for d in everyone]))
        stages_by_rank = {k: {'max': max(d.get(k, 0.0) for d in everyone), 'min': min(d.get(k, 0.0) for d in everyone),
                              'argmax_rank': int(np.argmax([d.get(k, 0.0) for d in everyone]))} for k in keys}

    def timed_alt(alt_enc):
        """The same K steps with another encoder object; returns (seconds over all ranks, profile snapshot)."""
        nonlocal enc
        enc_main, enc = enc, alt_enc
        try:
            step()
            _lib.profile_enable(True)
            _lib.profile_reset()
            fence()
            t_0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            secs = time.perf_counter() - t_0
            snap = _lib.profile_snapshot()
            _lib.profile_enable(False)
        finally:
            enc = enc_main
        if world > 1:
            tt = torch.tensor([secs], dtype=torch.float64, device=backend.collective_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            secs = float(tt.item())
        return secs, snap

    # The same workload once more per opt-in encoder mode; reported under "alt*", never `value`.  Each leg leaves the
    # first rows of its encodings behind so that the CPU sample below puts an error figure on it as well.
    alt = alt_f16 = alt_sparse = None
    alt_samples = {}
    keep_rows = min(args.cpu_sample_cells, hi - lo)

    def gemm16_leg(dtype, products, note):
        leg_enc = cc.HipEncoder(args.latent, args.genes, args.arch, seed=4321, batch_rows=args.batch_rows or None, dtype=dtype)
        secs, snap = timed_alt(leg_enc)
        a1 = snap['gemm_l1']
        executed = products * a1['work'] / (a1['ms'] * 1e-3) / 1e12 if a1['ms'] > 0 else 0.0
        leg = {'dtype': dtype, 'value': args.cells * args.steps / secs, 'unit': 'cells/s',
               'ms_per_step': secs / args.steps * 1e3,
               'layer1_algorithmic_tflops': executed / products, 'layer1_executed_fp16_tflops': executed,
               'roofline': {'kernel': f'gemm16_kernel (layer 1, {products} fp16 product(s) per algorithmic product)',
                            'bound': 'mfma', 'achieved': executed, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': executed / PEAK_F16_MFMA_TFLOPS, 'traffic': None,
                            'avg_launch_ms': a1['ms'] / max(a1['launches'], 1),   # a product whose last round of tiles is at most half full is two kernels (whole rounds + half-height tail)
                            'flop_per_launch': products * a1['work'] / max(a1['launches'], 1)},
               'fp32_fallback_passes': leg_enc.fp32_fallback_passes(),
               'stages_rank0': {k: {'launches': v['launches'], 'ms_per_step': v['ms'] / args.steps}
                                for k, v in snap.items() if v['launches']},
               'note': note}
        return leg, host_enc[:keep_rows].copy()

    # (N > 1: the side legs run only on request — a rank failing inside one would leave the others in a collective)
    if args.dtype == 'f32' and not args.sparse and not args.no_alt and (world == 1 or args.alt_multi):
        alt, alt_samples['alt'] = gemm16_leg(
            'f16x2', 2, 'encoder layers 1-2 as v_mfma_f32_16x16x32_f16 on W = W_hi + W_lo (and A = A_hi + A_lo for the sigmoid '
                        'activations), fp32 accumulate; same 1e-4 parity bound as f32; not the headline')
        alt_f16, alt_samples['alt_f16'] = gemm16_leg(
            'f16', 1, 'the arithmetic BASELINE configs[4] names: operands of layers 1-2 rounded to fp16 once, fp32 accumulate '
                      '(v_mfma_f32_16x16x32_f16); parity bound 1e-4 against the float64 evaluation of the rounded operands '
                      '(tests/test_gpu_encoder_f16.py, test_gpu_configs.py), NOT against the fp32 encoder; not the headline')
        # ... and with the sparse first layer: exact fp32 FMAs over the non-zeros only (csrc/spmm.hip)
        sp_enc = cc.HipEncoder(args.latent, args.genes, args.arch, seed=4321, sparse=True)
        sp_elapsed, sp_prof = timed_alt(sp_enc)
        alt_samples['alt_sparse'] = host_enc[:keep_rows].copy()
        s1 = sp_prof['spmm_l1']
        nz_tflops = s1['work'] / (s1['ms'] * 1e-3) / 1e12 if s1['ms'] > 0 else 0.0
        # the binding resource is LDS read bandwidth: every non-zero costs each of its layer-1 column panels one
        # ds_read_b128 wave instruction = 1 KiB (the 4 weights of that k-row per lane), i.e. 4 B of LDS per FMA
        lds_tbs = nz_tflops / 2.0 * 4.0
        alt_sparse = {'dtype': 'f32', 'value': args.cells * args.steps / sp_elapsed, 'unit': 'cells/s',
                      'ms_per_step': sp_elapsed / args.steps * 1e3,
                      'roofline': {'kernel': 'spmm_bias_act_kernel<1> (layer 1 over the non-zeros, fp32 vector FMAs)',
                                   'bound': 'valu', 'achieved': nz_tflops, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': nz_tflops / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
                                   'avg_launch_ms': s1['ms'] / max(s1['launches'], 1),
                                   'flop_per_launch': s1['work'] / max(s1['launches'], 1),
                                   'fp32_fma_tflops': nz_tflops,
                                   'lds_bytes_per_launch': s1['work'] / 2.0 * 4.0 / max(s1['launches'], 1),
                                   'lds_read_tbs': lds_tbs, 'lds_read_peak_tbs': PEAK_LDS_TBS,
                                   'frac_of_lds_read_peak': lds_tbs / PEAK_LDS_TBS,
                                   'dense_equivalent_tflops': (2.0 * (hi - lo) * args.genes * enc.layers[1].units * args.steps
                                                               / (s1['ms'] * 1e-3) / 1e12) if s1['ms'] > 0 else None,
                                   'note': 'achieved = FMA FLOP over the non-zeros / kernel time against the fp32 vector peak '
                                           '(= the fp32 MFMA figure on this part).  Round 4 probes (tools/experiments/README.md): '
                                           'dropping half of the LDS weight reads changes nothing, dropping one of the five '
                                           'vector instructions per entry 3-12 %: vector issue binds, not LDS bandwidth (the '
                                           'LDS figures stay as secondary fields); 2 packed FMAs among 5 vector instructions '
                                           'per entry cap a kernel of this shape at ~0.4 of the peak.'},
                      'stages_rank0': {k: {'launches': v['launches'], 'ms_per_step': v['ms'] / args.steps}
                                       for k, v in sp_prof.items() if v['launches']},
                      'note': 'sparse-aware first layer (SURVEY 8 f4): x @ K1 and the raw-input rows of K2 as CSR x dense '
                              'products over the ~5% non-zeros, fp32 FMA on the vector ALUs; no gather, no dense copy of '
                              'the counts; same 1e-4 parity bound (tests/test_gpu_sparse.py, test_gpu_configs.py at this '
                              'size); opt-in, not the headline'}
        del sp_enc

    # Side numbers for the reference's actual call shapes (dense host input: BasicBiGan.encoding_prediction gets a
    # DataFrame, reference src/bigan_basic.py:28-30; keras predict batches of 32, src/__main__.py:44 batch_size 128).
    dense_host = None
    if world == 1 and not args.no_dense_host and (args.dense_host or not args.no_alt) and args.dtype == 'f32' and not args.sparse:
        nd = min(16384, hi - lo)
        ld = (args.genes + 31) // 32 * 32
        dense_t = torch.empty((nd, ld), dtype=torch.float32, pin_memory=True)
        _lib.check(_lib.lib().cc_gather_dense(local.device_handle(), None, 0, nd, ld, dense_t.data_ptr(), 0))
        pinned = dense_t[:, :args.genes].contiguous().pin_memory()
        pageable = pinned.numpy().copy()

        def rate(x, reps):
            enc.predict(x)
            t_0 = time.perf_counter()
            for _ in range(reps):
                enc.predict(x)
            return len(x) * reps / (time.perf_counter() - t_0)

        def latency_ms(rows):
            x = pageable[:rows]
            enc.predict(x)
            ts = []
            for _ in range(20):
                t_0 = time.perf_counter()
                enc.predict(x)
                ts.append(time.perf_counter() - t_0)
            return float(np.median(ts) * 1e3)

        lat32, lat128 = latency_ms(32), latency_ms(128)
        _lib.check(_lib.lib().cc_encoder_set_low_latency(enc._h, 1))      # split-K form for calls of <= 128 rows
        lat32_ll, lat128_ll = latency_ms(32), latency_ms(128)
        _lib.check(_lib.lib().cc_encoder_set_low_latency(enc._h, 0))
        dense_host = {'cells': nd, 'dense_host_cells_per_s': rate(pageable, 2),
                      'dense_host_pinned_cells_per_s': rate(pinned.numpy(), 2),
                      'pcie_bound_cells_per_s': PCIE_GBS * 1e9 / (4.0 * args.genes),
                      'latency_ms_batch32': lat32, 'latency_ms_batch128': lat128,
                      'latency_ms_batch32_low_latency': lat32_ll, 'latency_ms_batch128_low_latency': lat128_ll,
                      'note': 'cc_encode_dense: fp32 rows from host memory (pageable / pinned), copies of pass i+1 under the '
                              'kernels of pass i; bounded by the host link (spec 63 GB/s), never part of `value`'}
        del dense_t, pinned, pageable

    if rank == 0:
        traffic = None   # HBM-side bytes per launch, REPLAYED from the PMC passes committed under profiles/
        # (round 4: the layer-1 kernel of whole rounds is the instance <1, false>; <1, true> = its half-height tail tiles)
        dominant = ('spmm_bias_act_kernel<1>',) if args.sparse else ('gemm_bias_act_kernel<1, false>', 'gemm_bias_act_kernel<1>')
        traffic_profile = next((f for f in TRAFFIC_PROFILES if os.path.exists(os.path.join(ROOT, 'profiles', f))), None)
        try:
            kernels = json.load(open(os.path.join(ROOT, 'profiles', traffic_profile)))['kernels']
            pmc = next(v for name in dominant for k, v in kernels.items() if name in k)
            fetch = pmc['FETCH_SIZE']['mean_per_launch'] if isinstance(pmc['FETCH_SIZE'], dict) else pmc['FETCH_SIZE']
            write = pmc['WRITE_SIZE']['mean_per_launch'] if isinstance(pmc['WRITE_SIZE'], dict) else pmc['WRITE_SIZE']
            traffic = (2 * fetch + write) * 1024
        except Exception:
            pass
        g1 = prof['spmm_l1'] if args.sparse else prof['gemm_l1']
        avg_ms = g1['ms'] / max(g1['launches'], 1)
        flop_per_launch = g1['work'] / max(g1['launches'], 1)
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        stages = {k: {'launches': v['launches'], 'ms_per_step': v['ms'] / args.steps,
                      'rate': (v['work'] / (v['ms'] * 1e-3)) if v['ms'] > 0 else 0.0}
                  for k, v in prof.items() if v['launches']}
        w0 = int(args.genes * 0.1) if args.arch == 'cont' else 1000
        shape = f'x[B x {args.genes}] @ K1[{args.genes} x {w0}]'
        if args.dtype != 'f32':
            # fp16 matrix cores: `f16x2` executes two fp16 products per algorithmic product (W = W_hi + W_lo)
            mult = 2.0 if args.dtype == 'f16x2' else 1.0
            roofline = {'kernel': f'gemm16_kernel (encoder layer 1, {shape}, {args.dtype})', 'bound': 'mfma',
                        'achieved': achieved * mult, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': achieved * mult / PEAK_F16_MFMA_TFLOPS, 'traffic': None,
                        'avg_launch_ms': avg_ms, 'flop_per_launch': flop_per_launch * mult,
                        'note': 'executed fp16 products; the algorithmic rate is achieved / %g' % mult}
        else:
            roofline = {'kernel': ('spmm_bias_act_kernel<1> (encoder layer 1 over the non-zeros of x, fp32 vector FMAs)'
                                   if args.sparse else f'gemm_bias_act_kernel<1> (encoder layer 1, {shape})'),
                        'bound': 'valu' if args.sparse else 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS,
                        'traffic': None if args.sparse else traffic,
                        'traffic_note': f'REPLAYED, not measured in this run: bytes per launch of the dominant kernel, '
                                        f'2*FETCH_SIZE+WRITE_SIZE from separate rocprofv3 --pmc passes of the same command '
                                        f'(profiles/{traffic_profile}: one 8192-row launch, the shape of the timed passes); algorithmic minimum 1.45e9 for the dense kernel',
                        'avg_launch_ms': avg_ms, 'flop_per_launch': flop_per_launch}
        line = {
            'metric': baseline_metric(), 'value': args.cells * args.steps / elapsed,
            'unit': 'cells/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{workload_name(args, world)}: synthetic {args.cells} cells x {args.genes} genes (5% dense '
                                   f'integer counts), {args.arch} encoder latent={args.latent} (random Glorot weights), '
                                   f'exact k={args.k} brute-force kNN',
                       'cells': args.cells, 'genes': args.genes, 'latent': args.latent, 'k': args.k,
                       'arch': args.arch, 'parallelism': f'cells sharded over {world} GPU(s), 1 all-gather/step' + (' (library exchange: cc_knn_l2_sharded)' if native is not None else ''),
                       'knn_fallback_queries_rank0': int(_lib.lib().cc_knn_last_fallback_count())},
            'roofline': roofline,
            'stages_rank0': stages,
        }
        if parity is not None:
            line['parity_check'] = parity
        if stages_by_rank is not None:
            line['stages_by_rank'] = stages_by_rank
            line['allgather_ms'] = stages_by_rank.get('host_allgather_ms')
        else:
            line['host_stage_ms_rank0'] = main_host_ms
        if alt is not None:
            line['alt'] = alt
        if alt_f16 is not None:
            line['alt_f16'] = alt_f16
        if alt_sparse is not None:
            line['alt_sparse'] = alt_sparse
        if dense_host is not None:
            line['dense_host'] = dense_host
        ksel = [prof[c] for c in ('knn_pack', 'knn_select', 'knn_rerank', 'knn_fallback')]
        knn_ms = sum(v['ms'] for v in ksel) / args.steps
        if knn_ms > 0:
            # algorithmic work of the search: 2 * queries * candidates * d.  Executed on the fp16 matrix
            # cores: ONE fp16 product per pair (hi x hi) + the norm step = ND + 1 MFMA K-steps of 16, by the collect
            # pass over all candidates and by the sample pass over every S-th block (S as the library chose it)
            knn_flop = 2.0 * (hi - lo) * args.cells * args.latent
            ktf = knn_flop / (knn_ms * 1e-3) / 1e12
            nd = -(-args.latent // 16)
            stat = ctypes.c_int64()
            _lib.check(_lib.lib().cc_knn_stat(None, 2, ctypes.byref(stat)))   # CC_KNN_STAT_SAMPLE_STRIDE of the last search
            stride = int(stat.value) or 8
            _lib.check(_lib.lib().cc_knn_stat(None, 3, ctypes.byref(stat)))   # CC_KNN_STAT_COLLECT_NORM_FORM: 2 = the collect's tile has no norm K-step
            collect_steps = nd if int(stat.value) == 2 else nd + 1
            executed = 16.0 * (collect_steps + (nd + 1) / stride) / args.latent
            line['roofline_knn'] = {'kernel': 'knn_sample16 + knn_collect16c + knn_merge + knn_rerank (per step)',
                                    'bound': 'mfma', 'achieved': ktf, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': ktf / PEAK_F16_MFMA_TFLOPS, 'executed_fp16_tflops': executed * ktf,
                                    'ms_per_step': knn_ms, 'flop_per_step': knn_flop, 'traffic': None,
                                    'note': 'achieved = algorithmic 2*q*c*d FLOP / stage time; executed = x%.3f (one fp16 '
                                            'product per pair in %d MFMA K-steps of 16: collect over all candidates%s, sample '
                                            'with the norm K-step over every %d-th block); parking and re-checking the passing '
                                            'pairs (VALU/LDS) shares the time'
                                            % (executed, collect_steps, ' with the norms through the accumulator' if collect_steps == nd else
                                               ' incl. the norm K-step', stride)}
        gather = prof['gather']
        if gather['ms'] > 0:
            gbs = gather['work'] / (gather['ms'] * 1e-3) / 1e9
            line['roofline_gather'] = {'kernel': 'csr_gather_kernel', 'bound': 'hbm', 'achieved': gbs,
                                       'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS}
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(args, enc, local, main_sample, alt_samples)
            for name, rel in base.pop('alt_max_rel').items():   # each leg carries its own error figure
                leg = {'alt': alt, 'alt_f16': alt_f16, 'alt_sparse': alt_sparse}.get(name)
                if leg is not None:
                    leg['max_rel_vs_cpu_sample'] = rel
                elif name == 'alt_f16_vs_f16_operand_oracle' and alt_f16 is not None:
                    alt_f16['max_rel_vs_f16_operand_oracle_sample'] = rel
            line['cpu_baseline'] = base
        print(json.dumps(line), flush=True)
    if native is not None:
        native.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity['ok']:
        raise SystemExit(f'bench.py rank {rank}: parity check FAILED: {json.dumps(parity)}')


if __name__ == '__main__':
    main()
