"""Instruction census of a gfx950 kernel from hipcc -S output: per basic block the number of MFMA / VALU / LDS / VMEM /
SALU instructions and the distribution of "fillers between consecutive MFMAs" (the guide's <= 5 hidden per 32-cycle gap).
usage: python tools/isa_census.py file.s [kernel-name-substring] [--blocks]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
    show_blocks = '--blocks' in sys.argv
    lines = open(path).read().split('\n')
    inside = False
    label = None
    blocks = []          # (label, [ops])
    for ln in lines:
        if re.match(r'^[A-Za-z_][\w$.]*:', ln) and not ln.startswith('.L'):
            name = ln.split(':')[0]
            inside = sub in name and not name.startswith('__')
            if inside:
                blocks.append((name, []))
            continue
        if not inside:
            continue
        if ln.startswith('.LBB') or ln.startswith('.Ltmp'):
            if ln.startswith('.LBB'):
                blocks.append((ln.split(':')[0], []))
            continue
        s = ln.strip()
        if not s or s.startswith((';', '.', '//')):
            continue
        op = s.split()[0]
        blocks[-1][1].append(op)
        if op == 's_endpgm':
            pass
    tot = Counter()
    gaps = Counter()
    mfma_kinds = Counter()
    for name, ops in blocks:
        c = Counter(classify(o) for o in ops)
        tot.update(c)
        run = None
        for o in ops:
            k = classify(o)
            if k == 'mfma':
                mfma_kinds[o] += 1
                if run is not None:
                    gaps[min(run, 12)] += 1
                run = 0
            elif run is not None and k != 'nop':
                run += 1
        if show_blocks and ops:
            print('%-12s %5d ops  %s' % (name[:12], len(ops), dict(c)))
    print('total', dict(tot))
    print('mfma kinds', dict(mfma_kinds))
    print('fillers between consecutive MFMAs (same block; 12 = 12 or more):')
    for k in sorted(gaps):
        print('  %2d: %d' % (k, gaps[k]))
    # cumulative: the instructions in runs > 5 that do not fit under the preceding MFMA
    over = sum((k - 5) * v for k, v in gaps.items() if k > 5)
    print('fillers beyond 5 per gap (lower bound, 12+ truncated):', over)


if __name__ == '__main__':
    main()
