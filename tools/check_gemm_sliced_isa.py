#!/usr/bin/env python3
"""Static check of csrc/gemm_sliced.hip's hand-managed memory instructions (hipcc cross-compiles without a GPU).
The kernel issues its activation loads and LDS reads as asm and places the waits itself; that is only sound while
the compiler leaves those registers alone.  Checked on the ISA of gemm_sliced_kernel:
  * no spills;
  * helper loop: exactly 8 `global_load_dwordx4` per period, into 8 distinct register tuples that are the SAME
    tuples in every period (each load refills the registers of the pair it follows: no rotation, hence no copy of
    a register whose load is still in flight), no `v_mov` reads those registers, and the first instruction that
    reads a loaded tuple comes after an `s_waitcnt vmcnt(6)`;
  * matrix loop: 28 `ds_read_b128` and 112 `v_mfma_i32_16x16x64_i8` per slab, every MFMA behind at least one
    `s_waitcnt lgkmcnt` that follows the read of its operands, 7 LDS-DMA pieces per slab and `vmcnt(7)` at the
    barrier.
usage: check_gemm_sliced_isa.py  -> exit code 0 / 1"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'l2hmc-qcd_amd', 'csrc')


def isa() -> str:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'gs.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                        '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-S', '--cuda-device-only', '-o', out,
                        os.path.join(CSRC, 'gemm_sliced.hip')], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def regs(tok):
    """v[a:b] or vN -> set of register numbers"""
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def operands(line):
    body = line.split(';')[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else '', []
    return parts[0], [t.strip().lstrip('-|').rstrip('|') for t in parts[1].split(',')]


def loops(lines):
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    out = []
    for n, l in enumerate(lines):
        m = re.search(r's_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if labels.get(t, 1 << 30) < n:
                out.append((labels[t], n))
    return out


def check(text: str):
    problems = []
    k = text[text.index('_ZN3l2q18gemm_sliced_kernelENS_6GsArgsEi:'):]
    k = k[:k.index('.Lfunc_end')]
    lines = k.split('\n')
    meta = text[text.index('.name:           _ZN3l2q18gemm_sliced_kernelENS_6GsArgsEi'):]
    spill = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', meta).group(1))
    vgpr = int(re.search(r'\.vgpr_count:\s+(\d+)', meta).group(1))
    if spill:
        problems.append(f'{spill} spilled VGPRs')
    helper = matrix = None
    for a, b in loops(lines):
        body = lines[a:b]
        nload = sum('global_load_dwordx4' in l and 'lds' not in l for l in body)
        nmfma = sum('v_mfma_i32_16x16x64_i8' in l for l in body)
        if nload == 8 and nmfma == 0:
            helper = (a, b)
        if nmfma >= 112 and (matrix is None or b - a < matrix[1] - matrix[0]):
            matrix = (a, b)
    if helper is None:
        problems.append('helper loop (8 loads per period) not found')
    else:
        body = lines[helper[0]:helper[1]]
        dests = []
        for l in body:
            op, ops_ = operands(l)
            if op == 'global_load_dwordx4':
                dests.append(frozenset(regs(ops_[0])))
        if len(set(dests)) != 8:
            problems.append(f'helper loads use {len(set(dests))} distinct tuples, not 8')
        loaded = set().union(*dests)
        waited = False
        for l in body:
            op, ops_ = operands(l)
            if op.startswith('s_waitcnt') and 'vmcnt(6)' in l:
                waited = True
            if op.startswith('v_mov') and any(regs(t) & loaded for t in ops_[1:]):
                problems.append('helper loop copies a loaded register: ' + l.strip())
            if op.startswith('v_') and not waited and any(regs(t) & loaded for t in ops_[1:]):
                problems.append('loaded register read before the first vmcnt(6): ' + l.strip())
    if matrix is None:
        problems.append('matrix loop (112 MFMAs per slab) not found')
    else:
        body = lines[matrix[0]:matrix[1]]
        nread = sum(l.strip().startswith('ds_read_b128') for l in body)
        nmfma = sum('v_mfma_i32_16x16x64_i8' in l for l in body)
        ndma = sum('global_load_lds_dwordx4' in l for l in body)
        if nread != 28 or ndma != 7:
            problems.append(f'matrix loop: {nread} ds_read_b128 (28), {ndma} LDS-DMA pieces (7)')
        if nmfma not in (112, 124):          # (the 12 tail MFMAs appear twice: after the barrier and at a range end)
            problems.append(f'matrix loop: {nmfma} MFMAs')
        if not any('vmcnt(7)' in l and 'lgkmcnt(0)' in l for l in body):
            problems.append('matrix loop: no vmcnt(7) lgkmcnt(0) in front of the barrier')
        # every MFMA operand tuple must have been waited for after it was read
        pending = {}            # register -> True while a ds_read into it has not been followed by a wait
        for l in body:
            op, ops_ = operands(l)
            if op == 'ds_read_b128':
                for r in sorted(regs(ops_[0])):
                    pending.pop(r, None)
                    pending[r] = True
            elif op.startswith('s_waitcnt') and 'lgkmcnt' in l:
                n = int(re.search(r'lgkmcnt\((\d+)\)', l).group(1))
                # in-order returns: all but the youngest n reads are complete (2 reads = 8 registers each... count tuples)
                order = [r for r in pending]
                keep = set(order[len(order) - 4 * n:]) if n else set()
                pending = {r: True for r in order if r in keep}
            elif op.startswith('v_mfma'):
                for t in ops_[1:3]:
                    if regs(t) & set(pending):
                        problems.append('MFMA reads a fragment whose ds_read has not been waited for: ' + l.strip())
                        break
    report = f'gemm_sliced_kernel: {vgpr} VGPRs, {spill} spills, helper loop {helper}, matrix loop {matrix}'
    return len(problems), report + ''.join('\n  ' + p for p in problems)


if __name__ == '__main__':
    bad, rep = check(isa())
    print(rep)
    sys.exit(1 if bad else 0)
