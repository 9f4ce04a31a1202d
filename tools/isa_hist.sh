#!/bin/bash
# usage: isa_hist.sh file.s kernel_mangled_prefix  -> instruction histogram of the kernel's largest loop body
f=$1; k=$2
awk "/^$k/,/s_endpgm/" $f > /tmp/_k.s
python3 - <<'PY'
import re,collections
lines=open('/tmp/_k.s').read().split('\n')
# find loops: label ... s_cbranch to earlier label
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB[0-9_]+):',l)
    if m: labels[m.group(1)]=i
best=None
for i,l in enumerate(lines):
    m=re.search(r's_cbranch_\w+\s+(\.LBB[0-9_]+)',l) or re.search(r's_branch\s+(\.LBB[0-9_]+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        span=(labels[m.group(1)],i)
        if best is None or span[1]-span[0]>best[1]-best[0]: best=span
print("loop lines",best)
c=collections.Counter()
for l in lines[best[0]:best[1]+1]:
    m=re.match(r'^\s+([a-z][a-z0-9_]+)',l)
    if m: c[m.group(1)]+=1
tot_v=sum(v for k,v in c.items() if k.startswith('v_'))
tot_ds=sum(v for k,v in c.items() if k.startswith('ds_'))
print("VALU",tot_v,"DS",tot_ds,"SALU",sum(v for k,v in c.items() if k.startswith('s_')),"scratch",sum(v for k,v in c.items() if k.startswith('scratch_')))
print(', '.join(f"{k}:{v}" for k,v in c.most_common(45)))
PY
