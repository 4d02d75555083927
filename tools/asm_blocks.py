#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing.

usage: asm_blocks.py listing.s mangled-kernel-substring [min_instructions]
Prints, for every basic block with at least min_instructions instructions, the counts by class (fp64 VALU, other VALU,
moves, DPP, quarter-rate, LDS, scalar, waitcnt / nop).  Development aid for the chains the pipeline kernels are bound by.
"""
import re, sys, collections

def classify(op):
    if op.startswith(('s_waitcnt', 's_nop')): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'flat_', 'scratch_', 'buffer_')): return 'vmem'
    if op.startswith(('v_rsq', 'v_rcp', 'v_sqrt', 'v_div', 'v_exp', 'v_log')): return 'quarter'
    if op.startswith(('v_accvgpr',)): return 'agpr'
    if op.startswith(('v_mov', 'v_cndmask', 'v_readlane', 'v_writelane', 'v_readfirstlane')): return 'mov'
    if '_f64' in op: return 'f64'
    if op.startswith('v_'): return 'valu'
    return 'other'

def main():
    path, key = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^[A-Za-z_0-9$]+:', l) and key in l.split(':')[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks, cur, name = [], collections.Counter(), lines[start]
    first = start
    for i in range(start + 1, end):
        l = lines[i].strip()
        if not l or l.startswith((';', '.p2align', '.amd', '.set')): continue
        if re.match(r'^\.?[A-Za-z_0-9$.]+:', l.split(';')[0].strip() or 'x'):
            blocks.append((name, first, cur)); cur = collections.Counter(); name = l; first = i; continue
        if l.startswith('.'): continue
        op = l.split()[0]
        c = classify(op)
        cur[c] += 1
        if 'dpp' in op or 'row_newbcast' in l or 'quad_perm' in l or 'row_sh' in l: cur['dpp'] += 1
    blocks.append((name, first, cur))
    for name, first, c in blocks:
        n = sum(v for k, v in c.items() if k != 'dpp')
        if n >= minn:
            print(f"{name:14s} line {first + 1:6d} n={n:5d} " + ' '.join(f"{k}={v}" for k, v in sorted(c.items())))

if __name__ == '__main__':
    main()
