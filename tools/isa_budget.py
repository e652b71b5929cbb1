#!/usr/bin/env python3
"""Static instruction budget of a kernel's main loop from `hipcc -S` output ("produce an ISA-level budget").

    python tools/isa_budget.py <file.s> <kernel name prefix> [--loop LABEL]

Prints, for the outermost loop of the kernel (or the loop whose header is LABEL) and for every inner loop inside it, the number of
VALU / packed-fp32 / transcendental / MFMA / LDS / VMEM / SALU / s_nop / s_waitcnt / barrier instructions.  Straight-line code of the
outer loop is executed once per group and wave; inner loops once per trip (the edge loop: once per edge of the lane's row)."""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_log', 'v_sin', 'v_cos')):
        return 'trans'
    if op.startswith('v_pk_'):
        return 'valu_pk'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op == 's_nop':
        return 's_nop'
    if op == 's_waitcnt':
        return 's_waitcnt'
    if op == 's_barrier':
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix))
    end = start
    while 's_endpgm' not in lines[end]:
        end += 1
    body = lines[start:end]
    # basic blocks and the loop each belongs to, from the compiler's comments (".LBBx_y:  ; in Loop: Header=BBx_z Depth=d" /
    # "=>This [Inner] Loop Header: Depth=d" / "Parent Loop BBx_z Depth=d" followed by "=>  This Inner Loop Header" lines)
    blocks = []                                   # (first line, last line, label, innermost loop header label, depth)
    cur = None
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):(.*)$', l)
        if m:
            if cur is not None:
                blocks.append((cur[0], i - 1) + cur[1:])
            lab, rest = m.group(1), m.group(2)
            j, txt = i + 1, rest
            while j < len(body) and body[j].lstrip().startswith(';'):      # continuation comment lines
                txt += ' ' + body[j]
                j += 1
            hm = re.search(r'This (?:Inner )?Loop Header: Depth=(\d+)', txt)
            im = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', txt)
            if hm:
                cur = (i, lab, lab[2:], int(hm.group(1)))
            elif im:
                cur = (i, lab, im.group(1), int(im.group(2)))
            else:
                cur = (i, lab, None, 0)
    if cur is not None:
        blocks.append((cur[0], len(body) - 1) + cur[1:])
    depth1 = collections.Counter(b[3] for b in blocks if b[4] == 1)
    outer_h = max(depth1, key=lambda h: sum(b[1] - b[0] for b in blocks if b[3] == h))
    # every deeper loop whose blocks lie between the outer loop's first and last block
    o_first = min(b[0] for b in blocks if b[3] == outer_h)
    o_last = max(b[1] for b in blocks if b[3] == outer_h)
    inner_heads = sorted({b[3] for b in blocks if b[4] >= 2 and o_first <= b[0] <= o_last}, key=lambda h: min(b[0] for b in blocks if b[3] == h))

    def count_blocks(sel):
        c = collections.Counter()
        for b in blocks:
            if sel(b):
                for j in range(b[0], b[1] + 1):
                    m = re.match(r'^\s+([a-z_0-9]+)', body[j])
                    if m:
                        c[classify(m.group(1))] += 1
        return c
    keys = ['valu', 'valu_pk', 'trans', 'mfma', 'lds', 'vmem', 'salu', 's_nop', 's_waitcnt', 'barrier']
    print('kernel %s\nouter loop header %s: lines %d..%d of the kernel body' % (prefix, outer_h, o_first, o_last))
    print('| region | ' + ' | '.join(keys) + ' | issue cycles (4 VALU, 16 trans, 8 MFMA) |')
    print('|---|' + '---|' * (len(keys) + 1))

    def row(name, c):
        cyc = 4 * (c['valu'] + c['valu_pk']) + 16 * c['trans'] + 8 * c['mfma']
        print('| %s | ' % name + ' | '.join(str(c[k]) for k in keys) + ' | %d |' % cyc)
    row('straight-line code of the group loop (once per group and wave)', count_blocks(lambda b: b[3] == outer_h and b[4] == 1))
    for h in inner_heads:
        lo = min(b[0] for b in blocks if b[3] == h)
        hi = max(b[1] for b in blocks if b[3] == h)
        row('inner loop %s (lines %d..%d), per trip' % (h, lo, hi), count_blocks(lambda b, h=h: b[3] == h))


if __name__ == '__main__':
    main()
