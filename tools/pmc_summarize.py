"""Summarise a rocprofv3 --pmc run (``*_counter_collection.csv``) per kernel: launches, mean duration and
the mean of every collected counter, as a markdown table.  Only anyloc kernels (or names matching
``--match``) are listed; the first ``--skip`` launches of each kernel are dropped as warm-up.

    python tools/pmc_summarize.py gpurun_out/pmc_fetch [--match gemm_h3] [--skip 1]
"""
import argparse
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(anyloc::\(anonymous namespace\)::|anyloc::)?([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    if not m:
        return name[:60]
    return (m.group(2) + (m.group(3) or ""))[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--match", default="anyloc")
    ap.add_argument("--skip", type=int, default=1)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {a.dir}")
    csv.field_size_limit(1 << 30)
    per = collections.OrderedDict()       # kernel -> dispatch id -> {counter: value, "_dur": ns}
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if a.match not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"]) + f" grid={r['Grid_Size']}"
                d = per.setdefault(k, collections.OrderedDict()).setdefault(r["Dispatch_Id"], {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                d["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                d["_vgpr"] = r["VGPR_Count"]
                d["_lds"] = r["LDS_Block_Size"]
    counters = sorted({c for k in per.values() for d in k.values() for c in d if not c.startswith("_")})
    print("| kernel | launches | VGPR | LDS | mean us | " + " | ".join(counters) + " |")
    print("|---|---|---|---|---|" + "---|" * len(counters))
    for k, disp in per.items():
        rows = list(disp.values())[a.skip:] or list(disp.values())
        n = len(rows)
        mean = lambda key: sum(r.get(key, 0.0) for r in rows) / n
        print(f"| `{k}` | {n} | {rows[0]['_vgpr']} | {rows[0]['_lds']} | {mean('_dur'):.1f} | " +
              " | ".join(f"{mean(c):.6g}" for c in counters) + " |")


if __name__ == "__main__":
    main()
