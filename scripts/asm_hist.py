"""Instruction histogram / loop body dump of one kernel in a hipcc -save-temps .s file.
usage: asm_hist.py file.s <kernel-name-substring> [dump]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if sys.argv[2] in l.split(':')[0] and ':' in l and not l.startswith(('.', '\t', ' ')))
end = next(i for i in range(start, len(lines)) if lines[i].strip() == 's_endpgm')
body = [l.strip() for l in lines[start + 1:end + 1]]
ins = [l for l in body if l and not l.startswith(('.', ';')) and not l.endswith(':')]
c = Counter(l.split()[0] for l in ins)
print(len(ins), 'instructions')
groups = Counter()
for k, v in c.items():
    g = 'valu' if k.startswith('v_') and not k.startswith('v_mfma') else 'mfma' if k.startswith('v_mfma') else 'salu' if k.startswith('s_') and not k.startswith(('s_load', 's_buffer', 's_waitcnt')) else 'smem' if k.startswith(('s_load', 's_buffer')) else 'wait' if k.startswith('s_waitcnt') else 'lds' if k.startswith('ds_') else 'vmem'
    groups[g] += v
print(dict(groups))
for k, v in c.most_common(45):
    print(f'{k:28s}{v}')
if len(sys.argv) > 3:
    print('\n'.join(lines[start:end + 1]))
