#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes (one directory per pass, csv output, <= 3 counters each) per kernel instantiation:

    for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" ...; do
        HESIC_NO_OVERLAP=1 rocprofv3 --pmc $set --kernel-include-regex "igemm_glds_kernel|sconv_n2w|sconv_w2n" --output-format csv \\
            -d gpurun_out/pmcsq/pN -- python profiles/scripts/forward_n.py hsic 3
    done
    python profiles/make_pmc_sq_json.py gpurun_out/pmcsq profiles/r02_pmc_sq.json

Per kernel: the mean of every counter per launch (summed over the chip's shader engines / SIMDs as rocprofv3 reports it) and a few
ratios.  Units (MI355X_MICROARCH.md, PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave,
SQ_BUSY_CYCLES counts cycles per shader engine (32 of them), SQ_VALU_MFMA_BUSY_CYCLES cycles per SIMD (1024)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    root, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "p*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            m = re.search(r"(\w+<[^>]*>|\w+)\(", name)
            k = (m.group(1) if m else name).replace(" ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {"launches": max(len(v) for v in cs.values()), "mean_per_launch": {c: round(x, 1) for c, x in sorted(mean.items())}}
        wc = mean.get("SQ_WAVE_CYCLES")
        if wc:
            e["share_of_wave_cycles"] = {n: round(mean[c] / wc, 4) for n, c in (
                ("issuing", "SQ_ACTIVE_INST_ANY"), ("waiting_to_issue", "SQ_WAIT_INST_ANY"), ("parked_waitcnt_barrier", "SQ_WAIT_ANY"),
                ("valu", "SQ_ACTIVE_INST_VALU"), ("lds", "SQ_ACTIVE_INST_LDS"), ("vmem", "SQ_ACTIVE_INST_VMEM"), ("waiting_on_lds", "SQ_WAIT_INST_LDS")) if c in mean}
        if "SQ_BUSY_CYCLES" in mean and "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
            e["kernel_kcycles"] = round(mean["SQ_BUSY_CYCLES"] / 32 / 1e3, 1)
            e["mfma_busy_share_of_kernel_cycles"] = round((mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (mean["SQ_BUSY_CYCLES"] / 32), 4)
        if mean.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in mean:       # SQ_INSTS_VALU counts the MFMAs too
            e["non_mfma_valu_per_mfma"] = round((mean["SQ_INSTS_VALU"] - mean["SQ_INSTS_MFMA"]) / mean["SQ_INSTS_MFMA"], 3)
            if "SQ_INSTS_SALU" in mean:
                e["salu_per_mfma"] = round(mean["SQ_INSTS_SALU"] / mean["SQ_INSTS_MFMA"], 3)
        if "SQ_LDS_IDX_ACTIVE" in mean and "SQ_LDS_BANK_CONFLICT" in mean and mean["SQ_LDS_IDX_ACTIVE"]:
            e["lds_bank_conflict_share_of_lds_cycles"] = round(mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"], 4)
        res[k] = e
    json.dump({"note": "rocprofv3 --pmc passes over HESIC_NO_OVERLAP=1 python profiles/scripts/forward_n.py hsic N (B=8, 512x512, bf16 maps, default analysis mode, one stream); "
                       "see the docstring of profiles/make_pmc_sq_json.py for the units", "kernels": res}, open(out, "w"), indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["mean_per_launch"].get("SQ_BUSY_CYCLES", 0))[:8]:
        print(k, json.dumps({x: e[x] for x in e if x != "mean_per_launch"}))
        print("   ", {c: v for c, v in e["mean_per_launch"].items() if "FIFO" in c or "CONFLICT" in c or "INST_CYCLES" in c or "LEVEL" in c or "COEXEC" in c or c in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE")})


if __name__ == "__main__":
    main()
