"""Build-time guarantees that need no GPU: the assembly checks of tools/check_kernel_asm.py (run by `make -C var_amd/csrc` on every
build) — no inline-asm instruction may touch an MFMA result before a compiler-padded reader (the round-2 attention hazard), and the
kernels with hand-counted vmcnt waits must not spill."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import check_kernel_asm as cka                      # noqa: E402

MFMA = '\tv_mfma_f32_32x32x2_f32 v[32:47], v1, v2, v[32:47]'


def _asm(body):
    return ['k:'] + body.splitlines() + ['\ts_endpgm']


def test_checker_flags_an_asm_first_reader_of_an_mfma_result():
    """the bug of round 2 in miniature: v_max3 from an asm statement right behind the score MFMAs"""
    bad = cka.check_hazards(_asm(MFMA + '\n\t;;#ASMSTART\n\tv_max3_f32 v145, v145, v36, v37\n\t;;#ASMEND'))
    assert len(bad) == 1 and 'v_max3_f32' in bad[0]
    # accumulators in AGPRs, asm reads the AGPR range directly
    bad = cka.check_hazards(_asm('\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]\n\t;;#ASMSTART\n\tv_accvgpr_read_b32 v9, a7\n\t;;#ASMEND'))
    assert len(bad) == 1
    # an s_nop that is too short does not help
    bad = cka.check_hazards(_asm(MFMA + '\n\ts_nop 7\n\t;;#ASMSTART\n\tv_max3_f32 v145, v145, v36, v37\n\t;;#ASMEND'))
    assert len(bad) == 1


def test_checker_accepts_compiler_padded_reads_and_long_enough_gaps():
    ok = MFMA + '\n\ts_nop 15\n\ts_nop 0\n\tv_max_f32_e32 v145, v32, v33\n\t;;#ASMSTART\n\tv_max3_f32 v145, v145, v36, v37\n\t;;#ASMEND'
    assert cka.check_hazards(_asm(ok)) == []
    # the accumulate chain itself (next MFMA takes the result as C) is not a hazard, and does not settle anything either
    chain = MFMA + '\n' + MFMA + '\n\tv_accvgpr_read_b32 v49, a47\n\tv_max_f32_e32 v1, v40, v41\n\t;;#ASMSTART\n\tv_max3_f32 v145, v1, v36, v37\n\t;;#ASMEND'
    assert cka.check_hazards(_asm(chain)) == []
    # 20 wait states issued behind the MFMA: settled by time
    gap = MFMA + '\n\ts_nop 15\n\ts_nop 3\n\t;;#ASMSTART\n\tv_max3_f32 v145, v145, v36, v37\n\t;;#ASMEND'
    assert cka.check_hazards(_asm(gap)) == []
    # asm that touches other registers is none of the checker's business
    other = MFMA + '\n\t;;#ASMSTART\n\ts_mov_b32 m0, s5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v100, s[2:3]\n\t;;#ASMEND'
    assert cka.check_hazards(_asm(other)) == []


def test_checker_follows_loop_back_edges():
    """an MFMA at the bottom of a loop body, an inline-asm reader of its result at the top of the next iteration: only visible across the branch"""
    loop = ('.LBB0_1:\n\t;;#ASMSTART\n\tv_max3_f32 v145, v145, v36, v37\n\t;;#ASMEND\n\tv_add_f32 v1, v1, v1\n\ts_nop 15\n\ts_nop 7\n'
            + MFMA + '\n\ts_cbranch_scc1 .LBB0_1')
    bad = cka.check_hazards(_asm(loop))
    assert len(bad) == 1 and 'through a branch' in bad[0]
    # the same loop with a compiler-generated first reader at the top is fine
    ok = loop.replace('.LBB0_1:\n', '.LBB0_1:\n\ts_nop 15\n\ts_nop 1\n\tv_max_f32_e32 v9, v36, v37\n', 1)
    assert cka.check_hazards(_asm(ok)) == []
    # a forward branch over the padding into an asm reader
    fwd = MFMA + '\n\ts_cbranch_vccz .LBB0_2\n\ts_nop 15\n\ts_nop 7\n.LBB0_2:\n\t;;#ASMSTART\n\tv_max3_f32 v145, v145, v36, v37\n\t;;#ASMEND'
    bad = cka.check_hazards(_asm(fwd))
    assert len(bad) == 1


def test_checker_inplace_asm_mfma_chain():
    """`v_mfma d, a, b, d` from inline asm (the accumulator pinned to one tuple, elem16.h): reading the previous MFMA's result as src C of the SAME tuple
    is the accumulate chain, not a hazard; an asm MFMA that takes a fresh result as its A / B operand, or as a different tuple, still is"""
    a = '\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]\n\t;;#ASMEND\n'
    assert cka.check_hazards(_asm(a + a + a)) == []
    bad = cka.check_hazards(_asm(a + '\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[12:15], v[8:11], v[4:7], v[12:15]\n\t;;#ASMEND'))
    assert len(bad) == 1
    bad = cka.check_hazards(_asm(a + '\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[12:15], v[0:3], v[4:7], v[8:11]\n\t;;#ASMEND'))
    assert len(bad) == 1
    # and a compiler-generated reader right behind an asm MFMA gets no padding from the compiler: it has to come >= 20 wait states later on its own
    # (this checker cannot see that: the first non-asm toucher settles an MFMA by definition) — documented limit; VH16_MFMA_16x16x32_INPLACE's users keep a
    # barrier and a waitcnt between the last MFMA and the epilogue


def test_checker_scratch_rules():
    meta = ['k1:                 ; @k1', '.LBB0_1:       ; =>This Inner Loop Header: Depth=1', '\tscratch_load_dword v1, off, off', '\ts_endpgm',
            'k2:                 ; @k2', '.LBB1_0:', '\tscratch_store_dword off, v1, off', '.LBB1_1:       ; =>This Inner Loop Header: Depth=1', '\tv_add_f32 v1, v1, v1', '\ts_endpgm',
            '    .name:           k1', '    .private_segment_fixed_size: 8', '    .sgpr_spill_count: 0', '    .vgpr_spill_count: 2',
            '    .name:           k2', '    .private_segment_fixed_size: 4', '    .sgpr_spill_count: 0', '    .vgpr_spill_count: 1',
            '    .name:           k3', '    .private_segment_fixed_size: 0', '    .sgpr_spill_count: 39', '    .vgpr_spill_count: 0']      # SGPRs in VGPR lanes: no memory
    assert len(cka.check_scratch(meta)) == 2
    assert len(cka.check_scratch(meta, outside_ok=['k2'])) == 1                        # k2 spills outside its loop only
    bad = cka.check_scratch(meta, outside_ok=['k1', 'k2'])
    assert len(bad) == 1 and 'inside a loop' in bad[0] and 'k1' in bad[0]
    assert cka.check_scratch(meta, outside_ok=['k2'], ok=['k1']) == []


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc (cross-compiles without a GPU)')
def test_shipped_kernels_pass_the_assembly_checks():
    """the check the Makefile runs on attn / attn16 / gemm16 / conv16 (device assembly from `hipcc -S --cuda-device-only`)"""
    out = subprocess.run(['make', '-C', os.path.join(ROOT, 'var_amd', 'csrc'), '-j4', 'asmcheck'], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    for name in ('attn', 'attn16', 'gemm16', 'conv16'):
        path = os.path.join(ROOT, 'var_amd', 'csrc', 'build', f'{name}.s')
        assert os.path.exists(path)
        lines = open(path).read().splitlines()
        assert cka.check_hazards(lines, path) == []
        assert sum(1 for ln in lines if 'v_mfma' in ln) > 0
