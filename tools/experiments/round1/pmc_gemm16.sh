#!/bin/bash
# GPU-box helper: SQ counters of the fp16 GEMM (f16x2 encoder, one 8192-row pass).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
root=$PWD
groups=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE")
i=0
for g in "${groups[@]}"; do
  out=$root/gpurun_out/pmc_gemm16/g$i
  rm -rf "$out"; mkdir -p "$out"
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -- python3 $root/bench.py --cells 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-alt --dtype ${DTYPE:-f16x2} >"$out/run.log" 2>&1) || { echo "group $i failed"; tail -5 "$out/run.log"; }
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_gemm16/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_gemm16/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
res = {}
for k, v in acc.items():
    if 'gemm' not in k: continue
    res[k] = {c: sum(x) / len(x) for c, x in v.items()}
    res[k]['ms'] = sum(dur[k]) / max(len(dur[k]), 1)
json.dump(res, open('gpurun_out/pmc_gemm16.json', 'w'), indent=1)
for k, v in res.items():
    if v['ms'] < 0.3: continue
    cyc = v.get('GRBM_GUI_ACTIVE', 0) / 8
    print(k[:44], 'ms=%.2f' % v['ms'], 'clk=%.2fGHz' % (cyc / v['ms'] / 1e6) if cyc else '', 'mfma_busy=%.2f' % (v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc)) if cyc else '',
          {c: ('%.3g' % x) for c, x in v.items() if c != 'ms'})
PY
