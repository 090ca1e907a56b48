#!/usr/bin/env python3
"""Static check of the kernels that hide tile loads from the compiler's wait bookkeeping (asm global_load + counted
s_waitcnt vmcnt(N), see gemm.hip): between such a load and the wait that covers it, no instruction may READ the destination
registers -- the compiler believes the value is there as soon as the asm statement has 'executed', so a register copy it
inserts at a branch or merge (seen once: gemm_bwd.hip, round 3) silently moves stale data.  The scan walks the control-flow
graph of every kernel in the ISA listing with the in-order queue of vector-memory operations as its state (counted waits
retire all but the youngest N) and reports every instruction that touches a register whose hidden load may still be in
flight on some path.

    python3 check_asm_loads.py [file.hip ...]      (default: every kernel file that uses the idiom)
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT = ['gemm.hip', 'gemm_split.hip', 'gemm_skinny.hip', 'gemm_tn.hip', 'gemm_bwd.hip', 'train_persist.hip', 'train_persist_bwd.hip', 'train_persist_top.hip', 'train_persist_topb.hip']
LOAD = re.compile(r'global_load_dword(x2|x3|x4)? v(?:\[(\d+):(\d+)\]|(\d+))')
REG = re.compile(r'v\[(\d+):(\d+)\]|\bv(\d+)\b')


def isa_of(path, tmp):
    out = os.path.join(tmp, os.path.basename(path) + '.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S',
                           '-I' + os.path.join(HERE, '..', '..', 'include'), path, '-o', out], stderr=subprocess.DEVNULL)
    return open(out).read().split('\n')


def functions(lines):
    """-> {name: [(line number, text)]} of the kernels in an ISA listing"""
    out, name = {}, None
    for no, raw in enumerate(lines, 1):
        t = raw.strip()
        m = re.match(r'^([A-Za-z_][\w$.]*):', t)
        if m and not t.startswith('.L'):
            name = m.group(1)
            out[name] = []
            continue
        if name is not None:
            out[name].append((no, t))
            if t.startswith('.size') or t.startswith('.Lfunc_end'):
                name = None
    return out


def blocks_of(body):
    """basic blocks: label -> instructions, label -> successor labels; the entry block is ''.  A block ends at a label and at
    every branch (the instructions behind a conditional branch form an anonymous block of their own: what is issued there must
    not be charged to the path that took the branch)."""
    blocks, order, cur, anon = {'': []}, [''], '', 0
    ends = {}                                   # label -> (targets, falls through)
    for no, t in body:
        m = re.match(r'^(\.LBB\w+):', t)
        if m:
            ends.setdefault(cur, ([], True))
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        blocks[cur].append((no, t))
        m = re.match(r'^s_(c?branch)\w*\s+(\.LBB\w+)', t)
        if m or t.startswith('s_endpgm'):
            ends[cur] = ([m.group(2)] if m else [], bool(m) and m.group(1) == 'cbranch')
            anon += 1
            cur = '%s+%d' % (order[-1].split('+')[0], anon)
            blocks[cur] = []
            order.append(cur)
    ends.setdefault(cur, ([], True))
    succ = {}
    for i, lab in enumerate(order):
        nxt = order[i + 1] if i + 1 < len(order) else None
        targets, falls = ends.get(lab, ([], True))
        succ[lab] = list(targets) + ([nxt] if falls and nxt else [])
    return blocks, succ


def run_block(instrs, state, bad):
    """state: tuple of queue entries, oldest first; an entry is ('x',) or (line, r0, r1) for a hidden load of v[r0:r1]"""
    q = list(state)
    in_asm = False
    for no, t in instrs:
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = LOAD.match(t)
        if m:
            if in_asm:
                r0, r1 = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
                q.append((no, r0, r1))
            else:
                q.append(('x',))
            # the address operand is a read like any other
            t = 'addr ' + t.split(',', 1)[1] if ',' in t else t
            ops = ['addr', t.split(None, 1)[1]]
        elif t.startswith(('global_store', 'global_atomic', 'buffer_', 'flat_', 'scratch_')):
            q.append(('x',))
            ops = t.split(None, 1)
        else:
            w = re.match(r'^s_waitcnt .*vmcnt\((\d+)\)', t)
            if w:
                n = int(w.group(1))
                q = q[len(q) - n:] if n else []
                continue
            if not t or t.startswith((';', '.', 's_')):
                continue
            ops = t.split(None, 1)
        hidden = [e for e in q if e[0] != 'x']
        if len(ops) < 2 or not hidden:
            continue
        for m2 in REG.finditer(ops[1]):
            lo, hi = (int(m2.group(3)), int(m2.group(3))) if m2.group(3) else (int(m2.group(1)), int(m2.group(2)))
            for e in hidden:
                if e[0] != no and lo <= e[2] and hi >= e[1]:
                    bad.add((no, 'v[%d:%d]' % (e[1], e[2]), e[0], t))
    return tuple(q[-64:])


def scan(lines):
    """-> sorted list of (line, registers, line of the load in flight, instruction): instructions that touch a register whose
    hidden load may still be in flight on some path"""
    bad = set()
    for name, body in functions(lines).items():
        blocks, succ = blocks_of(body)
        seen = {lab: set() for lab in blocks}
        work = [('', ())]
        while work:
            lab, state = work.pop()
            if state in seen[lab] or len(seen[lab]) > 400:
                continue
            seen[lab].add(state)
            out = run_block(blocks[lab], state, bad)
            for nxt in succ[lab]:
                if nxt in blocks:
                    work.append((nxt, out))
    return sorted(bad)


def main(argv):
    files = argv or DEFAULT
    total = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            path = f if os.path.isabs(f) else os.path.join(HERE, f)
            bad = scan(isa_of(path, tmp))
            total += len(bad)
            print('%-20s %s' % (os.path.basename(path), 'ok' if not bad else '%d reads of registers with a load in flight' % len(bad)))
            for no, regs, since, t in bad[:10]:
                print('    line %d touches %s (load at line %d still in flight): %s' % (no, regs, since, t))
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
