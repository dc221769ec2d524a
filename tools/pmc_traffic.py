"""Build profiles/pmc_traffic.json (read by bench.py for `roofline.traffic`) from two rocprofv3 --pmc passes over the
same ViT-g forward at the bench batch (tools/pmc_target_vit.py): one with FETCH_SIZE, one with WRITE_SIZE.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <gemm mode> [out.json]

Per launch of each block GEMM: bytes = 2 * FETCH_SIZE (KiB -> B; doubled because on gfx950 FETCH_SIZE tallies the
128-byte requests of 16-byte-per-lane streaming reads at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, against the
algorithmic bytes of that GEMM (operand images + output once)."""
import collections
import csv
import glob
import json
import os
import re
import sys

M, D, HID = 61 * 530, 1536, 4096
ALGO = {   # bytes one launch has to move at least: A image + W image (4 B per element as two fp16 planes) + output
    "vit_qkv_gemm": 4 * (M * D + 3 * D * D) + 4 * M * 3 * D,
    "vit_proj_gemm": 4 * (M * D + D * D) + 8 * M * D,            # + read-modify-write of the residual stream
    "vit_w12_gemm": 4 * (M * D + 2 * HID * D) + 4 * M * HID,
    "vit_fc2_gemm": 4 * (M * HID + D * HID) + 8 * M * D,
}


def per_kernel(d, counter):
    csv.field_size_limit(1 << 30)
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"] != counter or "gemm_" not in r["Kernel_Name"]:
                continue
            m = re.search(r"(gemm_[a-z0-9]+_kernel<[^>]*>)", r["Kernel_Name"])
            key = (m.group(1) if m else r["Kernel_Name"][:80], r["Grid_Size"])
            out[key].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return out


def main():
    fetch_dir, write_dir, mode = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    table = {}
    for key, vals in fetch.items():
        name, grid = key
        args = [a.strip() for a in name[name.index("<") + 1:-1].split(",")]
        epi = int(args[6]) if "h3" in name or "x6" in name else None
        wvals = write.get(key, [])
        groups = [(vals, wvals)]
        tags = None
        if epi in (0, 5):
            tags = ["vit_qkv_gemm"] if int(grid) > 1_000_000 else None
        elif epi in (3, 7, 8, 9):                   # SwiGLU epilogues (8 / 9: from transposed accumulators)
            tags = ["vit_w12_gemm"]
        elif epi == 2:          # proj and fc2 share template and grid: the shorter launches are proj (K = 1536 vs 4096)
            cut = sorted(v[1] for v in vals)[len(vals) // 2 - 1] * 1.4
            wcut = sorted(v[1] for v in wvals)[len(wvals) // 2 - 1] * 1.4 if wvals else 0
            groups = [([v for v in vals if v[1] <= cut], [v for v in wvals if v[1] <= wcut]),
                      ([v for v in vals if v[1] > cut], [v for v in wvals if v[1] > wcut])]
            tags = ["vit_proj_gemm", "vit_fc2_gemm"]
        if not tags:
            continue
        for tag, (fv, wv) in zip(tags, groups):
            if not fv:
                continue
            f_kib = sum(v[0] for v in fv) / len(fv)
            w_kib = sum(v[0] for v in wv) / len(wv) if wv else 0.0
            total = 2.0 * f_kib * 1024 + w_kib * 1024
            table[tag] = {"kernel": name, "launches_sampled": len(fv), "fetch_size_kib": round(f_kib, 1),
                          "write_size_kib": round(w_kib, 1), "bytes_per_launch": round(total),
                          "algorithmic_bytes": ALGO[tag], "refetch_ratio": round(total / ALGO[tag], 2),
                          "mean_us": round(sum(v[1] for v in fv) / len(fv), 1),
                          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on tools/pmc_target_vit.py (B=61, ViT-g, 3 blocks); "
                                    "bytes = 2 x FETCH_SIZE + WRITE_SIZE"}
    try:
        old = json.load(open(out_path))
    except (OSError, ValueError):
        old = {}
    old[mode] = table
    json.dump(old, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
