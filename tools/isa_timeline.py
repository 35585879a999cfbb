"""Compressed instruction timeline of one kernel from hipcc -S output: one character per instruction
(M mfma, v valu, l LDS read, s LDS write, d other LDS, G global/buffer load, T global store, . salu, n s_nop, J branch,
[..] s_waitcnt with its counters, |B| barrier), basic-block labels on their own lines.
usage: python tools/isa_timeline.py file.s kernel-name-substring"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split('\n')
    sub = sys.argv[2]
    start = [i for i, l in enumerate(lines) if re.match(r'^[A-Za-z_][\w$.]*:', l) and sub in l.split(':')[0]][0]
    end = [i for i, l in enumerate(lines) if i > start and '.end_amdhsa_kernel' in l][0]
    out = []
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((';', '//')):
            continue
        if t.startswith('.'):
            if re.match(r'^\.LBB', t):
                out.append('\n' + t.split(':')[0] + ': ')
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            c = 'M'
        elif op.startswith(('ds_read', 'ds_load')):
            c = 'l'
        elif op.startswith(('ds_write', 'ds_store')):
            c = 's'
        elif op.startswith('ds_'):
            c = 'd'
        elif op.startswith(('buffer_load', 'global_load')):
            c = 'G'
        elif op.startswith(('global_store', 'buffer_store')):
            c = 'T'
        elif op.startswith('s_waitcnt'):
            c = '[' + t.split(None, 1)[1].split(';')[0].replace(' ', '') + ']'
        elif op.startswith('s_barrier'):
            c = '|B|'
        elif op.startswith('s_nop'):
            c = 'n'
        elif op.startswith(('s_cbranch', 's_branch')):
            c = 'J'
        elif op.startswith('s_'):
            c = '.'
        elif op.startswith('v_'):
            c = 'v'
        else:
            c = '?'
        out.append(c)
    print(''.join(out))


if __name__ == '__main__':
    main()
