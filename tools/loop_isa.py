"""Instruction mix of the loops of one kernel in a hipcc -S listing: python tools/loop_isa.py /tmp/x.s <mangled-name-prefix>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
s = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]))
e = next(i for i in range(s, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[s:e]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r's_cbranch\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(1), 1e9) < i:
        a = labels[m.group(1)]
        ins = [x.strip().split()[0] for x in body[a:i + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        print(m.group(1), 'len', len(ins))
        if len(ins) > 60: print(Counter(ins).most_common(40))
for l in lines[e:e + 60]:
    if 'NumVgprs' in l or 'Occupancy' in l or 'ScratchSize' in l or 'NumSgprs' in l: print(l)
