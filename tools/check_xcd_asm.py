#!/usr/bin/env python3
"""Static check of the row-group-chain kernels of the XCD-pair recurrence (csrc/lstm_xcd.hip: k_lstm_fwd_pair_chains /
k_lstm_bwd_pair_chains).

Those kernels issue loads in one phase and consume them in a later one, with `s_waitcnt vmcnt(N)` counted by hand.  Two things
must hold on the FINAL ISA, whatever the compiler felt like doing, and this script checks both on a fresh `hipcc -S`:

  1. the landing registers a[176:245] are written by the inline-asm global loads ONLY, and read only by v_accvgpr_read /
     MFMA operands (a compiler that parks a temporary there would be overwritten by a returning load);
  2. between the first MFMA and the last MFMA / inline-asm load (= inside the time loop, in file order), every
     `s_waitcnt vmcnt` is one of ours (inside an ASMSTART / ASMEND block) -- except directly behind the slow path's
     compiler-issued `global_load_dword ... sc1` of the error flag: a compiler-inserted wait in the loop would be a
     vmcnt computed without the asm loads, i.e. far too strict, and would serialise the memory queue again.

Exit code 0 = ok.  Used by tests/test_abi.py::test_chain_kernels_keep_their_landing_registers (CPU box: hipcc
cross-compiles without a GPU).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'few-shot-music-generation_amd', 'csrc')
LO, HI = 176, 245


def landing_refs(text):
    """AGPR numbers in [LO, HI] that an instruction's operand text touches"""
    hits = []
    for m in re.finditer(r'\ba\[(\d+):(\d+)\]|\ba(\d+)\b', text):
        if m.group(3) is not None:
            lo = hi = int(m.group(3))
        else:
            lo, hi = int(m.group(1)), int(m.group(2))
        if hi >= LO and lo <= HI:
            hits.append((lo, hi))
    return hits


def kernels(asm):
    """name -> list of lines of every *_pair_chains kernel body"""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r'^(_ZN4fsmg\S*k_lstm_(?:fwd|bwd)_pair_chains\S*):', line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if '.end_amdhsa_kernel' in line or re.match(r'^\s*\.section', line):
                out[name] = cur
                cur = None
            else:
                cur.append(line)
    return out


def check(name, lines):
    problems = []
    in_asm = False
    touches = []
    for i, raw in enumerate(lines):
        line = raw.strip()
        if line.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if line.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not line or line.startswith(';') or line.startswith('.'):
            continue
        code = line.split(';')[0]
        if landing_refs(code):
            touches.append((i, in_asm, code.strip()))
    loads = [i for i, ia, l in touches if ia and l.startswith('global_load')]
    if not loads:
        return ['no inline-asm load into a[%d:%d] found (kernel changed?)' % (LO, HI)]
    for i, ia, l in touches:
        op = l.split()[0]
        if ia:
            if not op.startswith('global_load'):
                problems.append('line %d: asm block touches a landing register with %s' % (i, l))
        else:
            dst = l.split()[1].rstrip(',')
            if landing_refs(dst) and not op.startswith('v_accvgpr_read'):
                problems.append('line %d: compiler WRITES a landing register: %s' % (i, l))
            if not (op.startswith('v_accvgpr_read') or op.startswith('v_mfma')):
                problems.append('line %d: unexpected use of a landing register: %s' % (i, l))
    mfmas = [i for i, raw in enumerate(lines) if raw.strip().startswith('v_mfma')]
    if not mfmas:
        return problems + ['no MFMA found']
    first, last = min(mfmas), max(max(loads), max(mfmas))
    in_asm = False
    prev_op = ''
    for i in range(first, last):
        line = lines[i].strip()
        if line.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if line.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not line or line.startswith(';') or line.startswith('.'):
            continue
        if line.startswith('s_waitcnt') and 'vmcnt' in line and not in_asm:
            if not (prev_op.startswith('global_load_dword ') and 'sc1' in prev_op):
                problems.append('line %d: compiler-inserted "%s" inside the time loop (behind "%s")' % (i, line, prev_op))
        prev_op = line
    return problems


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'xcd.s')
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-S',
               os.path.join(CSRC, 'lstm_xcd.hip'), '-o', out]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if proc.returncode != 0:
            print(proc.stdout)
            return 2
        asm = open(out).read()
    ks = kernels(asm)
    if len(ks) < 6:
        print('expected 6 chain kernels, found %d' % len(ks))
        return 1
    bad = 0
    for name, lines in sorted(ks.items()):
        problems = check(name, lines)
        print('%s: %s' % (name, 'ok' if not problems else '%d problem(s)' % len(problems)))
        for p in problems:
            print('   ' + p)
        bad += len(problems)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
