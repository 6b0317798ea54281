#!/usr/bin/env python3
"""Instruction census of one kernel's innermost loop from hipcc -S output:  isa_census.py file.s <mangled-name-substring>"""
import re
import sys
from collections import Counter

L = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(L) if re.match(r"^_Z\S*" + re.escape(sys.argv[2]) + r"\S*:", l)][0]
end = next(i for i in range(start, len(L)) if L[i].startswith("\t.end_amdhsa_kernel") or "; -- End function" in L[i])
ins, labels = [], {}
for l in L[start + 1:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not t or t.startswith((".", ";")):
        continue
    ins.append(t)
back = [(i, t.split()[1]) for i, t in enumerate(ins) if t.startswith("s_cbranch") and t.split()[1] in labels and labels[t.split()[1]] <= i]
i1, lab = max(back, key=lambda b: b[0] - labels[b[1]])
loop = ins[labels[lab]:i1 + 1]
c = Counter()
for t in loop:
    op = t.split()[0]
    k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "waitcnt" if op.startswith("s_waitcnt") else
         "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else op)
    c[k] += 1
print(len(loop), dict(c))
print(Counter(t.split()[0] for t in loop if t.startswith("v_") and not t.startswith("v_mfma")).most_common(16))
for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy"):
    m = [re.search(r"; " + k + r": (\d+)", l) for l in L[end:end + 80]]
    print(k, [x.group(1) for x in m if x][:1])
