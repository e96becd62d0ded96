#!/usr/bin/env python3
"""Static check of gfx950 assembly: no VGPR that an in-flight inline-asm `ds_read` will write may be touched before
the `s_waitcnt lgkmcnt` that retires the read.

Why (round 4, the `-DCC_PP_NOMFMA` GPU fault of round 3): the hand-placed fragment reads of csrc/gemm16.hip are
`asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) ...)`.  hipcc believes the output is written AT the
statement; the hardware writes it when the LDS returns the data.  As long as the outputs are consumed (by the MFMAs
behind `s_waitcnt lgkmcnt(0)` + `sched_barrier`) the registers stay allocated until then.  The ablation build that
removed the MFMAs left the outputs dead: the register allocator gave all eight reads of a phase the same quad
v[14:17] and reused v[14:15] at once as the 64-bit address of the next `global_load_lds` — the LDS data landed on top
of the address and the request went to a wild location (memory fault, GPU core dump).  This script makes the product's
side of that argument checkable: it scans every kernel of an assembly file and reports any instruction that reads or
writes a register of a pending inline-asm ds_read.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S cellcomm_amd/csrc/gemm16.hip -o gemm16.s
  python3 tools/check_async_lds.py gemm16.s

Model: LDS operations return in order, so `lgkmcnt(N)` retires all but the N newest LDS/scalar-memory operations
(the loops checked here hold no scalar loads).  Control flow is followed in layout order (labels and branches do not
clear the pending set), which is conservative for the straight-line phases these loops consist of.
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
LGKM = re.compile(r'lgkmcnt\((\d+)\)')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(path):
    """[(kernel, inline ds_reads seen, [violation strings])] for every function of the file."""
    results = []
    kernel, in_asm, pending, seen, bad = None, False, [], 0, []
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(';')[0].strip() if not raw.lstrip().startswith(';;#') else raw.strip()
        if raw.startswith('_Z') and raw.rstrip().split(';')[0].rstrip().endswith(':'):
            if kernel is not None:
                results.append((kernel, seen, bad))
            kernel, in_asm, pending, seen, bad = raw.split(':')[0], False, [], 0, []
            continue
        if kernel is None or not line:
            continue
        if line.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if line.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if line.startswith('.') or line.endswith(':'):
            continue
        op = line.split()[0]
        if op == 's_endpgm':
            pending = []
            continue
        if op == 's_waitcnt':
            m = LGKM.search(line)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            elif 'lgkmcnt' not in line and 'vmcnt' not in line and 'expcnt' not in line:
                pending = []           # a raw immediate: treat as a full wait
            continue
        touched = regs_of(line)
        busy = set().union(*[p[0] for p in pending]) if pending else set()
        if op.startswith('ds_read') and in_asm:
            dst = regs_of(line.split(',')[0])
            src = touched - dst
            hit = src & busy              # an address register that a pending read will overwrite
            if hit:
                bad.append(f'{path}:{ln}: `{line}` reads v{sorted(hit)} under a pending inline-asm ds_read')
            # several pending reads into the same registers are only harmless if nothing consumes them: report
            if dst & busy:
                bad.append(f'{path}:{ln}: `{line}` targets v{sorted(dst & busy)}, still pending from an earlier inline-asm ds_read')
            pending.append((dst, ln))
            seen += 1
            continue
        if op.startswith('ds_') or op.startswith('s_load') or op.startswith('s_buffer_load'):
            pending.append((set(), ln))   # counts for lgkmcnt, owns no tracked register
        hit = touched & busy
        if hit:
            first = min(p[1] for p in pending if p[0] & hit)
            bad.append(f'{path}:{ln}: `{line}` touches v{sorted(hit)} while the inline-asm ds_read of line {first} is in flight')
    if kernel is not None:
        results.append((kernel, seen, bad))
    return results


def main():
    status = 0
    for path in sys.argv[1:]:
        for kernel, seen, bad in check(path):
            if seen == 0:
                continue
            print(f'{kernel}: {seen} inline-asm ds_read instructions, {len(bad)} violation(s)')
            for b in bad[:12]:
                print('   ', b)
            status |= 1 if bad else 0
    return status


if __name__ == '__main__':
    sys.exit(main())
