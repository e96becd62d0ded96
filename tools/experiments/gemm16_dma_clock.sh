#!/bin/bash
# GPU-box helper: clock and MFMA-pipe occupancy of the prototype kernels (one rocprofv3 counter pass).
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
root=$PWD
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/gemm16_dma.hip -o /tmp/gemm16_dma 2>/dev/null || exit 1
out=$root/gpurun_out/g16dma_pmc; rm -rf "$out"; mkdir -p "$out"
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$out" -- /tmp/gemm16_dma 16384 > "$out/run.log" 2>&1) || { tail -3 "$out/run.log"; exit 1; }
grep TFLOP "$out/run.log"
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/g16dma_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)): acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('gpurun_out/g16dma_pmc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}; ms = sum(dur[k]) / len(dur[k]); cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(k[:60], 'ms=%.3f clk=%.2f GHz mfma_busy=%.2f lds_active=%.2f wait=%.2f' % (ms, cyc / ms / 1e6, m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), m['SQ_ACTIVE_INST_LDS'] * 4 / (256 * cyc), m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']))
PY
