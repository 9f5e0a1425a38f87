"""Static check of the hand-placed MFMA hazards of the single-pass attention backward (svol_amd/csrc/attention_bf16.hip,
attn_bwd_sp_bf16).  Every MFMA of that kernel is inline asm, and hipcc pads nothing behind an asm statement: an MFMA result that a
NON-MFMA instruction reads (v_exp on the score tile, v_mul on dP, ds_write of the dQ partial, v_accvgpr_read of dK / dV) needs
NumPasses + 4 wait states on gfx950 (LLVM GCNHazardRecognizer: 12 for an 8-pass, 20 for a 16-pass XDL op).  The kernel keeps every such
consumer dozens of instructions behind its producer by construction; this test compiles the file to assembly (no GPU needed, ~10 s) and
measures the distance in the code hipcc really emitted, so a compiler or source change that moves a consumer up fails here, not as a
rare wrong gradient.  Also checked (round 5): the 2 wait states between a VALU write and the MFMA that reads the register as an
operand.  Not checked: distances across a loop's back edge — every step of these loops ends in s_waitcnt + s_barrier, far longer than
either window."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, 'svol_amd', 'csrc', 'attention_bf16.hip')
# Model: an XDL op of P passes needs P + 4 wait states before a non-MFMA reader (gfx950).  One instruction = one wait state, s_nop N =
# N + 1, and an intervening MFMA = P: the matrix pipe is in order and fully paced (one 32x32x16 16-bit MFMA per 32 cycles = 8 passes
# per SIMD, MI355X_MICROARCH.md), so when a later MFMA issues, every earlier one has finished its passes.  The 32x32x16 16-bit forms
# are documented as 8-pass; the check must hold under P = 16 as well.
PASSES = (8, 16)


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if c and os.path.exists(c):
            return c
    return None


def _regs(tok):
    """'v[64:79]' / 'v12' / 'a[0:15]' -> (file, set of indices); anything else -> None"""
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r'([va])(\d+)', tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None


def _parse(line):
    line = line.split(';')[0].strip()
    if not line or line.endswith(':') or line.startswith('.'):
        return None
    parts = line.split(None, 1)
    ops = [t.strip() for t in parts[1].split(',')] if len(parts) > 1 else []
    return parts[0], [t.split()[0] if t else t for t in ops]


def _kernel_lines(asm, name):
    i = asm.index(name + ':')
    j = asm.index('.end_amdhsa_kernel', i)
    return asm[i:j].split('\n')


def _mfma_loops(lines):
    """(first, last) line of every loop (backward branch) that holds MFMAs, outermost span per loop header"""
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    spans = {}
    for n, l in enumerate(lines):
        m = re.search(r's_cbranch\S*\s+(\.LBB\d+_\d+)', l)
        if m and labels.get(m.group(1), 1 << 30) < n:
            spans[labels[m.group(1)]] = max(spans.get(labels[m.group(1)], 0), n)
    out = [(a, b) for a, b in sorted(spans.items()) if sum('v_mfma' in l for l in lines[a:b]) >= 10]
    return [(a, b) for a, b in out if not any((a2, b2) != (a, b) and a <= a2 and b2 <= b for a2, b2 in out)]   # innermost only


def _wait_states(op, ops, passes):
    if op == 's_nop':
        return int(ops[0], 0) + 1
    if op.startswith('v_mfma'):
        return passes
    return 1


def _check(lines, passes):
    NEED = passes + 4
    insts = [(n, _parse(l)) for n, l in enumerate(lines)]
    insts = [(n, p) for n, p in insts if p]
    worst = None
    n_mfma = 0
    for k, (n, (op, ops)) in enumerate(insts):
        if not op.startswith('v_mfma'):
            continue
        n_mfma += 1
        dst = _regs(ops[0])
        assert dst, (op, ops)
        states = 0
        for n2, (op2, ops2) in insts[k + 1:]:
            if op2.startswith('s_cbranch') or op2 in ('s_branch', 's_endpgm', 's_barrier'):
                break          # a barrier costs far more than the hazard window; branches end the linear walk (loop bodies are long)
            touched = [_regs(t) for t in ops2]
            hit = any(t and t[0] == dst[0] and (t[1] & dst[1]) for t in touched)
            if hit:
                chain = op2.startswith('v_mfma') and len(ops2) >= 4 and _regs(ops2[3]) == dst and _regs(ops2[0]) == dst \
                    and not any(_regs(t) and _regs(t)[0] == dst[0] and (_regs(t)[1] & dst[1]) for t in ops2[1:3])
                if not chain:
                    if worst is None or states < worst[0]:
                        worst = (states, lines[n].strip(), lines[n2].strip())
                    assert states >= NEED, f'{lines[n2].strip()!r} reads the result of {lines[n].strip()!r} after {states} wait states'
                break          # (an accumulate chain restarts the walk from its own MFMA)
            states += _wait_states(op2, ops2, passes)
            if states > 4 * NEED:
                break
    return n_mfma, worst


def _check_valu_to_mfma(lines):
    """The second hand-managed hazard: a VALU-written register (v_cvt_pk of P / dS, v_exp, v_mul) read by an MFMA as its
    A, B or C operand needs 2 wait states in between; hipcc pads this only across statement boundaries it can see.  Returns (MFMAs
    checked, the smallest distance found to ANY in-window VALU producer, or None)."""
    insts = [(n, _parse(l)) for n, l in enumerate(lines)]
    insts = [(n, p) for n, p in insts if p]
    closest = None
    n_mfma = 0
    for k, (n, (op, ops)) in enumerate(insts):
        if not op.startswith('v_mfma'):
            continue
        n_mfma += 1
        srcs = [r for r in (_regs(t) for t in ops[1:4]) if r]
        states = 0
        for n2, (op2, ops2) in reversed(insts[max(0, k - 8):k]):
            if op2.startswith('s_cbranch') or op2 in ('s_branch', 's_barrier'):
                break
            if op2.startswith('v_') and not op2.startswith('v_mfma') and ops2:
                dst = _regs(ops2[0])
                if dst and any(s[0] == dst[0] and (s[1] & dst[1]) for s in srcs):
                    if closest is None or states < closest[0]:
                        closest = (states, lines[n2].strip(), lines[n].strip())
                    assert states >= 2, f'{lines[n].strip()!r} reads {lines[n2].strip()!r} after {states} wait states (2 needed)'
                    break
            states += _wait_states(op2, ops2, 8) if not op2.startswith('v_mfma') else 8
            if states >= 2:
                break
    return n_mfma, closest


@pytest.mark.parametrize('flag', [[], ['-DSVOL_H16_FP16']], ids=['bf16', 'fp16'])
def test_single_pass_backward_mfma_results_are_not_read_early(tmp_path, flag):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip('hipcc not found')
    from svol_amd import build
    out = str(tmp_path / 'attention.s')
    cmd = [hipcc] + build.COMMON + build.PER_FILE.get('attention_bf16.hip', []) + flag + ['--cuda-device-only', '-S', SRC, '-o', out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    asm = open(out).read()
    name = [m for m in re.findall(r'^(_Z\w*attn_bwd_sp_bf16\w*):', asm, flags=re.M)]
    assert name, 'attn_bwd_sp_bf16 not found in the assembly'
    lines = _kernel_lines(asm, name[0])
    for passes in PASSES:
        n_mfma, worst = _check(lines, passes)
        assert n_mfma >= 40                  # the loop body alone holds 40
        assert worst is not None and worst[0] >= passes + 4
    n2, closest = _check_valu_to_mfma(lines)     # VALU-written operand -> MFMA: at least 2 wait states (asserted inside)
    assert n2 >= 40
    body = '\n'.join(lines)
    # the whole point of the asm forms: no accumulator <-> vector register copies inside the loop, nothing spilled
    meta = asm[asm.index('amdhsa.kernels'):]
    k = meta[meta.index(name[0]):]
    assert int(re.search(r'\.private_segment_fixed_size:\s*(\d+)', k).group(1)) == 0
    assert int(re.search(r'\.vgpr_spill_count:\s*(\d+)', k).group(1)) == 0
    loops = _mfma_loops(lines)
    assert len(loops) == 4                   # one step loop per keys-per-wave variant (NKB = 1..4)
    for a, b in loops:                       # accumulators move between the register halves in prologue / epilogue only
        assert not any('v_accvgpr' in l or 'scratch_' in l for l in lines[a:b])
    assert 'cmpswap' not in body             # dQ leaves as global_atomic_add_f32, not a compare-and-swap loop
    assert body.count('global_atomic_add_f32') >= 4


@pytest.mark.parametrize('flag', [[], ['-DSVOL_H16_FP16']], ids=['bf16', 'fp16'])
def test_hand_placed_forward_keeps_its_hazard_distances(tmp_path, flag):
    """The round-5 fast forward (attn_fwd_bf16_fast2) is asm statements in source order too: the scores of the next block are
    exponentiated a whole block after their MFMAs, a conversion feeds the PV product four statements later.  Same two static checks
    as the backward, and nothing spilled."""
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip('hipcc not found')
    from svol_amd import build
    out = str(tmp_path / 'attention.s')
    cmd = [hipcc] + build.COMMON + build.PER_FILE.get('attention_bf16.hip', []) + flag + ['--cuda-device-only', '-S', SRC, '-o', out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    asm = open(out).read()
    name = [m for m in re.findall(r'^(_Z\w*attn_fwd_\w*fast2\w*):', asm, flags=re.M)]
    assert name, 'attn_fwd_bf16_fast2 not found in the assembly'
    lines = _kernel_lines(asm, name[0])
    # the prologue (anchor from key tile 0) uses the MFMA BUILTIN, which hipcc pads itself for the documented 8 passes; the hand-managed
    # part is the tile loop: checked under both pass models
    n_mfma, worst = _check(lines, 8)
    assert n_mfma >= 16 and worst is not None and worst[0] >= 12
    loops = _mfma_loops(lines)
    assert len(loops) == 1
    a, b = loops[0]
    for passes in PASSES:
        n_loop, worst = _check(lines[a:b], passes)
        assert n_loop >= 16 and (worst is None or worst[0] >= passes + 4)
    n2, _ = _check_valu_to_mfma(lines)
    assert n2 >= 16
    meta = asm[asm.index('amdhsa.kernels'):]
    k = meta[meta.index(name[0]):]
    assert int(re.search(r'\.private_segment_fixed_size:\s*(\d+)', k).group(1)) == 0
    assert int(re.search(r'\.vgpr_spill_count:\s*(\d+)', k).group(1)) == 0
    body = '\n'.join(lines)
    assert 'v_pk_add_f32' not in body and 'v_pk_mul_f32' not in body   # a packed fp32 instruction does not overlap an MFMA


@pytest.mark.parametrize('flag', [[], ['-DSVOL_H16_FP16']], ids=['bf16', 'fp16'])
def test_no_attention_kernel_spills(tmp_path, flag):
    """VERDICT r4 "What's weak 8": attn_bwd_dq_bf16_rot / _pre_masked compiled to 256 VGPRs with 12 - 16 bytes of scratch per lane (the
    64-bit row indices of the epilogue and a block coordinate left in a vector register by block_coords' divisions, carried across the
    key loop).  Every kernel of attention_bf16.hip, both 16-bit types: no scratch, no spilled VGPRs."""
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip('hipcc not found')
    from svol_amd import build
    out = str(tmp_path / 'attention.s')
    cmd = [hipcc] + build.COMMON + build.PER_FILE.get('attention_bf16.hip', []) + flag + ['--cuda-device-only', '-S', SRC, '-o', out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    meta = open(out).read()
    meta = meta[meta.index('amdhsa.kernels'):]
    kernels = meta.split('  - .agpr_count:')[1:]
    assert len(kernels) >= 20
    bad = {}
    for k in kernels:
        name = re.search(r'\.name:\s*(\S+)', k).group(1)
        spill = int(re.search(r'\.vgpr_spill_count:\s*(\d+)', k).group(1))
        scratch = int(re.search(r'\.private_segment_fixed_size:\s*(\d+)', k).group(1))
        if spill or scratch:
            bad[name] = (spill, scratch)
    assert not bad, bad


def _store_data_hazards(asm_text, min_states=2):
    """every `buffer_store_dwordx{3,4}` whose soffset is an SGPR: the instructions within `min_states` wait states behind it must not
    WRITE one of its data registers.  LLVM pads this hazard only for stores without an SGPR soffset (GCNHazardRecognizer::
    createsVALUHazard); on gfx950 an overwrite right behind an SGPR-offset store corrupted the wave's last lanes (round 6,
    svol_amd/csrc/gemm_ws_bf16.hip::store_b128_guarded).  Returns (stores seen, list of violations)."""
    lines = asm_text.split('\n')
    insts = [(n, _parse(l)) for n, l in enumerate(lines)]
    insts = [(n, p) for n, p in insts if p]
    seen, bad = 0, []
    for k, (n, (op, ops)) in enumerate(insts):
        if not re.fullmatch(r'buffer_store_dwordx[34]', op) or len(ops) < 4:
            continue
        if not re.fullmatch(r's\d+', ops[3]):       # soffset: an SGPR (an immediate / `off` form is the case LLVM handles itself)
            continue
        data = _regs(ops[0])
        if not data:
            continue
        seen += 1
        states = 0
        for n2, (op2, ops2) in insts[k + 1:]:
            if states >= min_states or op2.startswith('s_cbranch') or op2 in ('s_branch', 's_endpgm', 's_barrier'):
                break
            if op2.startswith('v_') and ops2:
                dst = _regs(ops2[0])
                if dst and dst[0] == data[0] and (dst[1] & data[1]):
                    bad.append((states, lines[n].strip(), lines[n2].strip()))
                    break
            states += _wait_states(op2, ops2, 8)
    return seen, bad


@pytest.mark.parametrize('flag', [[], ['-DSVOL_H16_FP16']], ids=['bf16', 'fp16'])
def test_buffer_store_data_registers_are_not_overwritten_behind_the_store(tmp_path, flag):
    """Round 6's compiler trap: the weight-stationary GEMM's epilogue stores (buffer_store_dwordx4 with an SGPR soffset) followed at
    once by a VALU write of their data registers — hipcc emits no wait states there, the hardware needs some.  Static check of the
    emitted code of the file that uses such stores, both operand types."""
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip('hipcc not found')
    from svol_amd import build
    src = os.path.join(REPO, 'svol_amd', 'csrc', 'gemm_ws_bf16.hip')
    out = str(tmp_path / 'ws.s')
    subprocess.check_call([hipcc] + build.COMMON + flag + ['--cuda-device-only', '-S', src, '-o', out], stderr=subprocess.DEVNULL)
    seen, bad = _store_data_hazards(open(out).read())
    assert seen >= 40, seen                   # ten pipelined kernels x (loop + tail) x 2 - 4 stores
    assert not bad, bad[:5]
