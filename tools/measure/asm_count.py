#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing (device only): VALU / SALU / LDS / VMEM / MFMA / scratch
per kernel and per loop body (label-to-backward-branch ranges), to compare encoder variants without a GPU.
usage: asm_count.py listing.s substring-of-mangled-name"""
import re, sys, collections

def klass(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'): return 'MFMA'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'WAIT'
    if op.startswith('s_'): return 'SALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith('scratch_') or (op.startswith('buffer_') and 'offen' in op): return 'SCRATCH'
    if op.startswith(('global_', 'flat_', 'buffer_')): return 'VMEM'
    return 'OTHER'

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    print(lines[start])
    ins = []          # (line index, op, text)
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = len(ins); continue
        t = l.strip()
        if not t or t.startswith((';', '.', '//')): continue
        op = t.split()[0]
        ins.append((i, op, t))
    tot = collections.Counter(klass(op) for _, op, _ in ins)
    print('kernel total:', dict(tot))
    # loops: backward branches
    loops = []
    for k, (_, op, t) in enumerate(ins):
        if op.startswith('s_cbranch') or op == 's_branch':
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                loops.append((labels[tgt], k, tgt))
    for a, b, tgt in sorted(loops, key=lambda x: x[0] - x[1])[:6]:
        c = collections.Counter(klass(op) for _, op, _ in ins[a:b + 1])
        print(f'loop {tgt}: {b - a + 1} instructions', dict(c))
    if len(sys.argv) > 3:
        for _, op, t in ins: print(t)

main()
