#!/usr/bin/env python3
"""Static check of the row-group-chain kernels of the XCD-pair recurrence (csrc/lstm_xcd.hip: k_lstm_fwd_pair_chains /
k_lstm_bwd_pair_chains) and of the bf16-split pair kernels (csrc/lstm_pair16.h: check_pair16 below).

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


def check_pair16(asm):
    """k_lstm_fwd_pair16 / k_lstm_bwd_pair16 (csrc/lstm_pair16.h): 256 weight registers per lane leave the compiler ~200 for everything
    else, and the forward kernel's streamed fetch rests on the ORDER hipcc gives its own loads and waits.  On the final ISA of every
    product instantiation:
      1. nothing is spilled (no scratch_ instruction; a spill in the time loop is a memory operation in front of a poll);
      2. a step is 192 v_mfma_f32_16x16x32_bf16 (six products x eight k steps x four column tiles / two k steps x sixteen tiles);
      3. forward: the 24 streamed fragment loads (global_load_dwordx4 without sc1, in one run) are consumed progressively -- the first
         compiler-inserted wait behind them still leaves >= 18 loads in flight (hipcc used to cluster them by base register: 11)."""
    problems, found = [], 0
    for m in re.finditer(r'^(_ZN4fsmg\S*k_lstm_(fwd|bwd)_pair16ILi(\d)ELb0E\S*):', asm, re.M):
        name, direction = m.group(1), m.group(2)
        body = asm[m.end():asm.index('.Lfunc_end', m.end())]
        lines = [l.strip() for l in body.splitlines()]
        code = [l for l in lines if l and not l.startswith((';', '.'))]
        found += 1
        tag = 'k_lstm_%s_pair16<%s>' % (direction, m.group(3))
        if any(l.startswith('scratch_') for l in code):
            problems.append('%s: spills (scratch_ instructions)' % tag)
        n_mfma = sum(1 for l in code if l.startswith('v_mfma_f32_16x16x32_bf16'))
        if n_mfma != 192:
            problems.append('%s: %d MFMAs, expected 192' % (tag, n_mfma))
        if direction == 'fwd':
            idx = [i for i, l in enumerate(code) if l.startswith('global_load_dwordx4') and 'sc1' not in l]
            runs, cur = [], []
            for i in idx:
                if cur and i - cur[-1] > 12:
                    runs.append(cur); cur = []
                cur.append(i)
            if cur:
                runs.append(cur)
            runs = [r for r in runs if len(r) == 24]
            if len(runs) != 1:
                problems.append('%s: the run of 24 streamed fragment loads was not found' % tag)
                continue
            first_wait = next((l for l in code[runs[0][-1] + 1:] if l.startswith('s_waitcnt') and 'vmcnt' in l), None)
            n = int(re.search(r'vmcnt\((\d+)\)', first_wait).group(1)) if first_wait else -1
            if n < 18:
                problems.append('%s: the first wait behind the streamed loads is "%s": the fetch is not consumed k step by k step' % (tag, first_wait))
    if found != 8:
        problems.append('expected 8 product instantiations of the pair16 kernels, found %d' % found)
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
    p16 = check_pair16(asm)
    print('k_lstm_*_pair16: %s' % ('ok' if not p16 else '%d problem(s)' % len(p16)))
    for p in p16:
        print('   ' + p)
    bad += len(p16)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
