"""Static checks on the BUILT gfx950 assembly of the product kernels (CPU: hipcc cross-compiles without a GPU).

Round 3's `-DCC_PP_NOMFMA` ablation of the fp16 ping-pong K loop faulted a GPU.  Round 4 found the cause in the ISA
(tools/experiments/README.md): the inline-asm `ds_read_b128` of csrc/gemm16.hip have "=v" outputs that the compiler
believes written at the statement; left dead they were re-allocated as the address of the next LDS-DMA request while the
read was still in flight.  The product keeps every output live until a consumer behind `s_waitcnt lgkmcnt(0)`;
tools/check_async_lds.py proves it on the assembly, and this test keeps proving it for every later edit of the file."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
HIPCC = '/opt/rocm/bin/hipcc'


def assemble(src, out, flags=()):
    subprocess.check_call([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S',
                           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'cellcomm_amd', 'csrc'),
                           *flags, src, '-o', out], stderr=subprocess.DEVNULL)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_no_register_of_a_pending_inline_asm_lds_read_is_touched(tmp_path):
    import check_async_lds
    src = os.path.join(ROOT, 'cellcomm_amd', 'csrc', 'gemm16.hip')
    asm = str(tmp_path / 'gemm16.s')
    assemble(src, asm)
    res = [r for r in check_async_lds.check(asm) if r[1]]
    assert len(res) >= 4                                         # the ping-pong kernels (1 and 2 planes, 2 output modes)
    assert sum(r[1] for r in res) >= 80
    for kernel, seen, bad in res:
        assert not bad, (kernel, bad[:3])

    # the checker does see the fault of round 3: the same source with the MFMAs of the ping-pong loop removed
    text = open(src).read()
    body = re.search(r'auto mfma16 = \[&\]\(int hf, const h8 \(&af\)\[4\], const h8 \(&bf\)\[4\]\) \{.*?\n  \};', text, re.S).group(0)
    gutted = re.sub(r'#pragma unroll\n    for \(int mt = 0; mt < 4; \+\+mt\)\n#pragma unroll\n      for \(int nt = 0; nt < 4; \+\+nt\)\n'
                    r'        acc\[4 \* hf \+ mt\]\[nt\] = [^\n]*\n', '', body)
    assert gutted != body
    variant = tmp_path / 'gemm16_nomfma.hip'
    variant.write_text(text.replace(body, gutted, 1))            # first occurrence = gemm16_segment_pp
    asm2 = str(tmp_path / 'gemm16_nomfma.s')
    assemble(str(variant), asm2)
    flagged = [r for r in check_async_lds.check(asm2) if r[2]]
    assert flagged and any('global_load_lds' in b for r in flagged for b in r[2])


def test_the_product_sources_have_no_result_destroying_switches():
    # `make EXTRA=-DCC_PP_NODMA` must build the same bits as `make` — the macros are gone from the product
    for name in os.listdir(os.path.join(ROOT, 'cellcomm_amd', 'csrc')):
        if name.endswith(('.hip', '.h', '.cpp', '.inc')):
            text = open(os.path.join(ROOT, 'cellcomm_amd', 'csrc', name)).read()
            for macro in ('CC_PP_NODMA', 'CC_PP_NOA', 'CC_PP_NOB', 'CC_PP_NOREAD', 'CC_PP_SAMEK', 'CC_PP_NOPRIO',
                          'CC_G16_NOFENCE', 'CC_G16_PP', 'CC_GEMM_FAKE'):
                assert not re.search(r'#\s*if.*\b' + macro, text), (name, macro)
    # (the experiment kit tools/experiments/g16_ablations.patch carries those switches; it applies to gemm16.hip as of commit
    # 8cf6e37 — the file before the A16T operand layout — which is where the ablations were measured)
    assert os.path.exists(os.path.join(ROOT, 'tools', 'experiments', 'g16_ablations.patch'))
