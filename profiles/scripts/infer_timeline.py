#!/usr/bin/env python3
"""One eval forward out of a rocprofv3 kernel trace (single stream, HESIC_NO_OVERLAP=1): the launches in order with durations, and
per-(kernel, grid) totals.

    HESIC_NO_OVERLAP=1 rocprofv3 --kernel-trace -d /tmp/p -o t --output-format csv -- python profiles/scripts/forward_n.py hsic 6
    python profiles/scripts/infer_timeline.py /tmp/p [--full]
A forward is delimited by the first kernel of the step (the conv1 + GDN launch on x1)."""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "sum_sq_diff_kernel" in r["Kernel_Name"] or "rd_sums_kernel" in r["Kernel_Name"]]
    # the metrics reductions close a step (one or two launches per step): take the launches of the last step
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 6
    per = max(1, len(marks) // nsteps)
    a, b = marks[-1 - per], marks[-1]
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in seg)
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"kernels {len(seg)}  span {(t1 - t0) / 1e3:.1f} us  sum {sum(map(dur, seg)) / 1e3:.1f} us")
    clean = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n))
    agg = defaultdict(lambda: [0, 0])
    for r in seg:
        k = (clean(r["Kernel_Name"])[:80], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
        agg[k][0] += 1
        agg[k][1] += dur(r)
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{t / 1e3:9.1f} us  x{n:3d}  avg {t / n / 1e3:7.1f}  blocks {k[1]:>7d}  {k[0]}")
    if "--full" in sys.argv:
        for r in seg:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {dur(r) / 1e3:7.1f} b{int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):>6d} {clean(r['Kernel_Name'])[:90]}")


if __name__ == "__main__":
    main()
