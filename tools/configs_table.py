#!/usr/bin/env python3
"""GPU-box helper: every BASELINE.json config shape that fits one GPU, as ONE tracked file.

  python3 tools/configs_table.py <out.json>        (e.g. profiles/r03_configs.json)

Each row is a fresh `bench.py` process (10 steps after 2 warm-ups; 3 after 1 for the rows above 100k cells; no CPU baseline, no alt legs) reduced to the fields the
DESIGN.md table quotes, plus — for the two sharded configs — one rank's kNN call against ALL candidates of the
config (tools/knn_bench.py: synthetic encodings already in HBM), which the one-GPU bench line cannot contain.
Nothing here feeds bench.py's `value`.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BENCH_ROWS = [
    ('configs[1] 10k x 20k, latent 32 (encode-only in BASELINE.json; the bench also builds the kNN graph)',
     ['--cells', '10000', '--genes', '20000', '--latent', '32']),
    ('configs[2] 100k x 30k, latent 64, k = 15: dense fp32 (the headline arithmetic)', []),
    ('configs[2], sparse first layer', ['--sparse']),
    ('configs[2], f16x2', ['--dtype', 'f16x2']),
    ('configs[2], f16', ['--dtype', 'f16']),
    ('configs[2] shapes, classify encoder', ['--arch', 'cls']),
    ("configs[3]: one rank's 125k of 1M cells (encode + kNN among its own rows)", ['--cells', '125000']),
    ("configs[4]: one rank's 62.5k of 500k x 33k cells, fp16 operands, k = 50 (encode + kNN among its own rows)",
     ['--cells', '62500', '--genes', '33000', '--k', '50', '--dtype', 'f16']),
    # the two 8-GPU configs at FULL size on ONE GPU: the N = 1 denominators of their ">= 6x at 8 GPUs" target (round 4)
    ('configs[3] at full size on one GPU: 1M x 30k, latent 64, k = 15 (one encode call of 1M rows, one 1M x 1M search per step)',
     ['--cells', '1000000']),
    ('configs[4] at full size on one GPU: 500k x 33k, fp16 operands, k = 50 (one 500k x 500k search per step)',
     ['--cells', '500000', '--genes', '33000', '--k', '50', '--dtype', 'f16']),
]
KNN_ROWS = [
    ("configs[3]: one rank's search, 125 000 queries x 1 000 000 candidates, d = 64, k = 15", ['1000000', '64', '15', '125000', '3']),
    ("configs[4]: one rank's search, 62 500 queries x 500 000 candidates, d = 64, k = 50", ['500000', '64', '50', '62500', '3']),
    ('configs[2] search alone, 100 000 x 100 000, d = 64, k = 15', ['100000', '64', '15', '100000', '5']),
]


def bench_row(label, flags):
    # (short steps take 10 timed steps: over 3, one hiccup of a few ms in a 10 ms step moved configs[1]'s row by 17 %)
    long_step = '--cells' in flags and int(flags[flags.index('--cells') + 1]) > 100000
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3' if long_step else '10', '--warmup', '1' if long_step else '2',
           '--no-cpu-baseline', '--no-alt'] + flags
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if proc.returncode != 0:
        return {'label': label, 'flags': flags, 'error': proc.stderr[-800:]}
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{')][-1])
    return {'label': label, 'flags': flags, 'workload': d['config']['workload'], 'dtype': d['dtype'],
            'cells_per_s': d['value'], 'ms_per_step': d['ms_per_step'],
            'dominant_kernel': d['roofline']['kernel'], 'roofline_frac': d['roofline']['frac'],
            'roofline_achieved': d['roofline']['achieved'], 'roofline_unit': d['roofline']['unit'],
            'stages_ms_per_step': {k: round(v['ms_per_step'], 3) for k, v in d['stages_rank0'].items()}}


def knn_row(label, args):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'knn_bench.py')] + args, capture_output=True,
                          text=True, timeout=600)
    if proc.returncode != 0:
        return {'label': label, 'args': args, 'error': proc.stderr[-800:]}
    line = proc.stdout.strip().splitlines()[-1]
    ms = float(re.search(r'([0-9.]+) ms/call', line).group(1))
    stages = json.loads(re.search(r'(\{.*\})', line).group(1).replace("'", '"'))
    return {'label': label, 'args': dict(zip(('n', 'd', 'k', 'nq', 'reps'), map(int, args))), 'ms_per_call': ms,
            'stages_ms_per_call': stages, 'fallback_queries': int(line.rsplit('fallback', 1)[1])}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'configs.json')
    rows = []
    for label, flags in BENCH_ROWS:
        rows.append(bench_row(label, flags))
        print(json.dumps(rows[-1])[:300], flush=True)
    knn = []
    for label, args in KNN_ROWS:
        knn.append(knn_row(label, args))
        print(json.dumps(knn[-1])[:300], flush=True)
    with open(out, 'w') as f:
        json.dump({'made_by': 'tools/configs_table.py', 'bench_rows': rows, 'knn_rows': knn}, f, indent=1)
    print('wrote', out)


if __name__ == '__main__':
    main()
