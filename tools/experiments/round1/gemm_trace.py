#!/usr/bin/env python3
"""Diagnostic (needs a build with EXTRA=-DCC_GEMM_TRACE): when do the workgroups of one layer-1 GEMM launch
start / reach K/2 / finish, per XCD and CU?  Answers whether co-resident tiles advance through K in step."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cellcomm_amd as cc
from cellcomm_amd import _lib, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
_lib.ensure_init(0)
m, _ = synth.synth_csr_device(rows, 30000, seed=1)
enc = cc.HipEncoder(64, 30000, 'cont', seed=2)
out = torch.empty((rows, 64), device='cuda')
for _ in range(2):
    enc.predict_to_device(m, out.data_ptr(), 0, rows)
torch.cuda.synchronize()
nwg = -(-rows // 128) * 24
buf = np.zeros((nwg, 4), dtype=np.uint64)
fn = _lib.lib().cc_debug_gemm_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data, nwg) == 0
t = buf[:, :3].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0                      # wall_clock64 ticks at 100 MHz
hw = (buf[:, 3] & 0xffffffff).astype(np.int64)
xcc = (buf[:, 3] >> 32).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print(f'{nwg} workgroups; kernel span {us[:, 2].max():.0f} us; durations min/med/max '
      f'{np.min(us[:, 2] - us[:, 0]):.0f}/{np.median(us[:, 2] - us[:, 0]):.0f}/{np.max(us[:, 2] - us[:, 0]):.0f} us')
bid = np.arange(nwg)
print('blockIdx%8 == XCC_ID for', float(np.mean((bid & 7) == xcc)), 'of the workgroups')
for x in range(8):
    sel = xcc == x
    st = np.sort(us[sel, 0])
    first = st < st[0] + 0.5 * np.median(us[:, 2] - us[:, 0])       # the first round of residents
    n1 = int(first.sum())
    mid = np.sort(us[sel, 1])[:n1]
    print(f'xcd {x}: {sel.sum()} wgs, first round {n1}: start spread {st[n1 - 1] - st[0]:.1f} us '
          f'(p10-p90 {np.percentile(st[:n1], 10) - st[0]:.1f}-{np.percentile(st[:n1], 90) - st[0]:.1f}), '
          f'K/2 spread {mid[-1] - mid[0]:.1f} us (p10-p90 {np.percentile(mid, 90) - np.percentile(mid, 10):.1f}), '
          f'distinct (se,sh,cu) {len(set(zip(se[sel], sh[sel], cu[sel])))}')
kstep_us = np.median(us[:, 2] - us[:, 0]) / 938
print(f'one K-step of a workgroup ~ {kstep_us:.2f} us')
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', f'gemm_trace_{rows}.npy'), buf)
# the workgroups dispatched to the XCD of blockIdx%8 == 0, in dispatch order: (CU, start, K/2, end) in us
sel = np.where((bid & 7) == 0)[0]
cuid = se * 32 + sh * 16 + cu
for j, b in enumerate(sel[:200]):
    print(f'j={j:3d} cu={cuid[b]:3d} start {us[b, 0]:8.1f} half {us[b, 1]:8.1f} end {us[b, 2]:8.1f}')
