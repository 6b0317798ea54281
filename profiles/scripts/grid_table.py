#!/usr/bin/env python3
"""Per (kernel, grid) table of a rocprofv3 kernel trace: launches, blocks, average duration and total time -- to read how full the
block slots of the chip are (256 CUs x the blocks a CU holds) launch by launch.
    rocprofv3 --kernel-trace -d /tmp/p -o t --output-format csv -- python profiles/scripts/train_step.py --size 512 --only g --steps 10
    python profiles/scripts/grid_table.py /tmp/p [min_total_us]"""
import collections
import csv
import glob
import re
import sys

f = (glob.glob(sys.argv[1] + '/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[0]
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)
    wg = max(1, int(r['Workgroup_Size_X']))
    key = (n[:70], int(r['Grid_Size_X']) // wg, wg)
    agg[key][0] += 1
    agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot:.0f} us over {sum(v[0] for v in agg.values())} launches")
for (n, blocks, wg), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t >= lim:
        print(f"{n:70s} blocks {blocks:6d} x {wg:4d}  n {c:5d}  avg {t / c:8.1f} us  total {t:9.0f} us  {100 * t / tot:5.1f} %")
