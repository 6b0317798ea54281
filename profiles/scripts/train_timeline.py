#!/usr/bin/env python3
"""One graphed training step out of a rocprofv3 kernel trace: per (kernel, grid) totals, span, busy and idle time, optional full timeline.

    rocprofv3 --kernel-trace -d /tmp/p -o t --output-format csv -- python profiles/scripts/train_step.py --size 512 --only g --steps 6
    python profiles/scripts/train_timeline.py /tmp/p [--full]
A step is delimited by the adam_update_kernel launches (two per step, the last kernels of the step).
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adam_update_kernel" in r["Kernel_Name"]]
    # the step's two Adam launches are adjacent in time; the step ends with the second
    ends = [ad[i] for i in range(1, len(ad), 2)]
    a, b = ends[-2], ends[-1]
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in seg)
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, (cs, ce) = 0, iv[0]
    for s_, e_ in iv[1:]:
        if s_ > ce:
            busy += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    busy += ce - cs
    print(f"kernels {len(seg)}  span {(t1 - t0) / 1e3:.1f} us  sum {sum(map(dur, seg)) / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {(t1 - t0 - busy) / 1e3:.1f} us")
    clean = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n))
    agg = defaultdict(lambda: [0, 0])
    for r in seg:
        k = (clean(r["Kernel_Name"])[:70], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        agg[k][0] += 1
        agg[k][1] += dur(r)
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{t / 1e3:9.1f} us  x{n:3d}  avg {t / n / 1e3:7.1f}  blocks {k[1]:>7d} x{k[2]} x{k[3]}  {k[0]}")
    if "--full" in sys.argv:
        for r in seg:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} {dur(r) / 1e3:7.1f} q{r.get('Queue_Id', '?')} "
                  f"b{int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):>7d} {clean(r['Kernel_Name'])[:70]}")


if __name__ == "__main__":
    main()
