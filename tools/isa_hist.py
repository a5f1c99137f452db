"""Instruction histogram of one kernel in a hipcc -S listing: python tools/isa_hist.py file.s <substring of the mangled name> [loop]
With 'loop': print the listing of the basic blocks between the labels given after it instead."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E') or (l.startswith('_Z') and key in l and ':' in l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
c = Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l[0] in ';.' or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
print(sum(c.values()), 'instructions')
for k, v in c.most_common(70):
    print(f'{k:28s}{v}')
