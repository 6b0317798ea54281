#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output) of one HESIC forward into
profiles/pmc_igemm.json, the per-launch HBM-side traffic that bench.py reports as roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --kernel-include-regex igemm_glds_kernel --output-format csv -d gpurun_out/pmc_f -- python profiles/scripts/forward_n.py hsic 4   (with HESIC_NO_OVERLAP=1)
    rocprofv3 --pmc WRITE_SIZE ...                                                          -d gpurun_out/pmc_w -- ...
    python profiles/make_pmc_json.py gpurun_out/pmc_f gpurun_out/pmc_w hsic_bf16_b8_512

Units / corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
counts 64 B for every 128-B request of a wide (16 B/lane) coalesced read, so the read side is doubled.  Infinity-Cache hits
are included (fabric-side counter): this is traffic leaving the XCD L2s, an upper bound on HBM bytes.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            m = re.search(r"igemm_glds_kernel<([^>]*)>", r["Kernel_Name"])
            if m:
                out[m.group(1).replace(" ", "")].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            m = re.search(r"igemm_tr4_kernel<([^>]*)>", r["Kernel_Name"])      # phase-fused transposed form: template argument = GDN mode
            if m:
                out["tr4:" + m.group(1).replace(" ", "")].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return out


def main():
    fdir, wdir, key = sys.argv[1:4]
    F, W = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    res = {}
    for inst in F:
        fr, wr = sorted(F[inst]), sorted(W.get(inst, []))
        n = min(len(fr), len(wr))
        rd = sum(v for _, v in fr[:n]) * 1024 * 2 / n
        wt = sum(v for _, v in wr[:n]) * 1024 / n
        res[inst] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wt, "hbm_bytes_per_launch": rd + wt}
    # bench.py names fused instantiations "...,gdn": merge the GDN=1 / GDN=2 variants of one tile shape
    merged = {}
    for inst, v in res.items():
        p = inst.split(",")
        if inst.startswith("tr4:"):
            name = f"igemm_tr4_kernel<{'gdn' if inst[4:] != '0' else 'plain'}>"
            m = merged.setdefault(name, {"launches": 0, "bytes": 0.0})
            m["launches"] += v["launches"]
            m["bytes"] += v["hbm_bytes_per_launch"] * v["launches"]
            continue
        hl = len(p) > 7 and p[7] == "1"
        name = f"igemm_glds_kernel<{p[0]},{p[1]},{p[2]}{',gdn' if p[4] != '0' else ''}>" + (" hilo" if hl else (" hilo-gdn-out" if p[4] in ("3", "4") else ""))
        m = merged.setdefault(name, {"launches": 0, "bytes": 0.0})
        m["launches"] += v["launches"]
        m["bytes"] += v["hbm_bytes_per_launch"] * v["launches"]
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_igemm.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data["_commit"] = os.environ.get("PMC_COMMIT", data.get("_commit"))
    data["_collected"] = os.environ.get("PMC_COLLECTED", data.get("_collected"))
    data[key] = {"per_kernel": {k: {"launches": v["launches"], "hbm_bytes_per_launch": round(v["bytes"] / v["launches"])} for k, v in merged.items()},
                 "raw_instantiations": res,
                 "note": "FETCH_SIZE x2 (gfx950 128-B request correction) + WRITE_SIZE, KiB -> bytes; fabric-side, includes Infinity-Cache hits"}
    json.dump(data, open(out_path, "w"), indent=1)
    print(json.dumps(data[key]["per_kernel"], indent=1))


if __name__ == "__main__":
    main()
