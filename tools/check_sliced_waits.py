#!/usr/bin/env python3
"""Static check of csrc/heads_sliced.hip's hand-counted waits: compile to ISA and verify, for every template
instance, that each `s_waitcnt vmcnt(N)` in front of an `s_barrier` has at least N vector-memory instructions
between the last LDS-DMA piece of its period and itself (LDS-DMA, loads and stores retire in order, so the DMA
is then complete when the barrier is passed).  The compiler is free to add memory instructions (spills) --
that only makes the wait stricter -- but must not move the operand loads of `fetch` above the DMA.
usage: check_sliced_waits.py  -> exit code 0 / 1"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'l2hmc-qcd_amd', 'csrc')


def isa() -> str:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'hs.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                        '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-S', '--cuda-device-only', '-o', out,
                        os.path.join(CSRC, 'heads_sliced.hip')], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def check(text: str):
    kern, cur = {}, None
    for line in text.split('\n'):
        m = re.match(r'^(_ZN3l2q19heads_sliced_kernel\S+):', line)
        if m:
            cur = m.group(1)
            kern[cur] = []
        elif cur is not None:
            kern[cur].append(line)
            if line.strip().startswith('.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
                cur = None
    bad, report = 0, []
    for k, lines in kern.items():
        ops = []
        for line in lines:
            t = line.strip().split()
            if not t:
                continue
            op = t[0]
            if op.startswith('global_load_lds'):
                ops.append('D')
            elif op.startswith(('global_load', 'global_store', 'scratch_', 'buffer_')):
                ops.append('V')
            elif op == 's_barrier':
                ops.append('B')
            elif op == 's_waitcnt' and 'vmcnt' in line:
                ops.append(('W', int(re.search(r'vmcnt\((\d+)\)', line).group(1))))
        since = None
        for i, o in enumerate(ops):
            if o == 'D':
                since = 0
            elif o == 'V' and since is not None:
                since += 1
            elif isinstance(o, tuple) and i + 1 < len(ops) and ops[i + 1] == 'B':
                if since is not None and o[1] > since:
                    bad += 1
                    report.append(f'{k}: vmcnt({o[1]}) with only {since} memory instructions after the DMA')
                since = None
        report.append(f'{k[28:60]}: {ops.count("B")} barriers, {ops.count("D")} LDS-DMA pieces')
    return len(kern), bad, report


if __name__ == '__main__':
    n, bad, rep = check(isa())
    print('\n'.join(rep))
    print(f'{n} kernels, {bad} violations')
    sys.exit(1 if bad or n != 12 else 0)
