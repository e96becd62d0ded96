#!/usr/bin/env python3
"""GPU-box helper: SQ / TCC counters of one command's kernels, one rocprofv3 pass per counter group
(--pmc with --kernel-trace only, as the pool requires), summarised per kernel to a JSON file.

  python3 tools/pmc_run.py <out.json> <kernel-substring>[,<substring>...] [--groups sq,tcc] -- <program> [args...]

<program> must be the program itself (python3 or a binary), never a shell wrapper.  Per kernel the summary holds
the mean counter value per launch, the mean launch time, the clock the chip held (GRBM_GUI_ACTIVE / 8 / time)
and the usual ratios (vector / matrix / LDS pipe busy, share of wave cycles spent waiting).
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

GROUPS = {
    'sq': ['SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE',
           'SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA GRBM_GUI_ACTIVE',
           'SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE'],
    'tcc': ['FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum TCC_MISS_sum'],
}


def main():
    argv = sys.argv[1:]
    if '--' not in argv or len(argv) < 4:
        raise SystemExit(__doc__)
    cut = argv.index('--')
    head, cmd = argv[:cut], argv[cut + 1:]
    out_json, subs = head[0], head[1].split(',')
    groups = ['sq']
    if '--groups' in head:
        groups = head[head.index('--groups') + 1].split(',')
    scratch = os.path.join('/tmp', 'pmc_run_' + os.path.basename(out_json).replace('.json', ''))
    shutil.rmtree(scratch, ignore_errors=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    env = dict(os.environ, TMPDIR='/tmp')
    npass = 0
    for g in groups:
        for counters in GROUPS[g]:
            out = os.path.join(scratch, f'p{npass}')
            npass += 1
            os.makedirs(out, exist_ok=True)
            full = ['rocprofv3', '--pmc'] + counters.split() + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--'] + cmd
            with open(os.path.join(out, 'run.log'), 'w') as log:
                rc = subprocess.call(full, cwd='/tmp', env=env, stdout=log, stderr=subprocess.STDOUT, timeout=900)
            if rc != 0:
                print(f'pass "{counters}" failed with {rc}:', open(os.path.join(out, 'run.log')).read()[-1500:])
                raise SystemExit(1)
            for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
            for f in glob.glob(os.path.join(out, '**', '*kernel_trace.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    dur[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
            print(f'pass {npass} ({counters}) done', flush=True)
    res = {}
    for k, v in acc.items():
        if not any(s in k for s in subs):
            continue
        r = {c: sum(x) / len(x) for c, x in v.items()}
        r['launches_per_pass'] = len(dur[k]) // max(npass, 1)
        r['ms'] = sum(dur[k]) / max(len(dur[k]), 1)
        cyc = r.get('GRBM_GUI_ACTIVE', 0) / 8
        if cyc and r['ms']:
            r['clock_ghz'] = cyc / r['ms'] / 1e6
            simd_cycles = 1024 * cyc          # 256 CUs x 4 SIMDs
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in r:
                r['mfma_pipe_busy'] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles
            if 'SQ_ACTIVE_INST_VALU' in r:
                r['valu_busy'] = 4 * r['SQ_ACTIVE_INST_VALU'] / simd_cycles        # quad-cycles
            if 'SQ_LDS_IDX_ACTIVE' in r:
                r['lds_busy'] = r['SQ_LDS_IDX_ACTIVE'] / (256 * cyc)
        if r.get('SQ_WAVE_CYCLES'):
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY'):
                if c in r:
                    r[c + '_share'] = r[c] / r['SQ_WAVE_CYCLES']
        if 'TCC_HIT_sum' in r and 'TCC_MISS_sum' in r:
            r['l2_hit'] = r['TCC_HIT_sum'] / (r['TCC_HIT_sum'] + r['TCC_MISS_sum'])
        if 'FETCH_SIZE' in r:
            r['hbm_side_bytes'] = (2 * r['FETCH_SIZE'] + r.get('WRITE_SIZE', 0)) * 1024   # guide: FETCH_SIZE counts half
        res[k] = r
    os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
    json.dump({'command': cmd, 'kernels': res}, open(out_json, 'w'), indent=1)
    for k, r in res.items():
        print(k[:70], {c: (float('%.4g' % x) if isinstance(x, float) else x) for c, x in r.items()
                       if c in ('ms', 'clock_ghz', 'mfma_pipe_busy', 'valu_busy', 'lds_busy', 'SQ_WAIT_ANY_share',
                                'SQ_WAIT_INST_ANY_share', 'SQ_WAIT_INST_LDS_share', 'SQ_ACTIVE_INST_ANY_share', 'l2_hit',
                                'hbm_side_bytes', 'launches_per_pass')})
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
