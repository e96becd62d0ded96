import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from cellcomm_amd import _lib
from cellcomm_amd.knn import knn_graph_device
_lib.ensure_init(0)
n, d, k = 100000, 64, 15
x = torch.rand((n, d), device='cuda'); torch.cuda.synchronize()
pi = torch.empty((n, k), dtype=torch.int32, pin_memory=True); pd = torch.empty((n, k), dtype=torch.float32, pin_memory=True)
out = (pi.numpy(), pd.numpy())
for label, o in (('pinned', out), ('pageable', None)):
    knn_graph_device(x.data_ptr(), n, d, k, 0, n, out=o)
    for prof in (False, True):
        _lib.profile_enable(prof); _lib.profile_reset()
        t0 = time.perf_counter()
        for _ in range(20):
            knn_graph_device(x.data_ptr(), n, d, k, 0, n, out=o)
        dt = (time.perf_counter() - t0) / 20 * 1e3
        p = _lib.profile_snapshot() if prof else {}
        print(label, 'profile', prof, f'{dt:.3f} ms/call', {kk: round(v['ms'] / 20, 3) for kk, v in p.items() if v['launches']})
_lib.profile_enable(False)
