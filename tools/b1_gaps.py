"""Where a one-image forward's time goes, from a rocprofv3 kernel trace of tools/b1_trace_target.py:
    python tools/b1_gaps.py <..._kernel_trace.csv> [forwards=12] > profiles/r04_b1_kernel_trace_gaps.md
The trace holds one row per dispatch with start / end timestamps (ns).  The last `forwards` forwards are cut out by
their launch count (every forward of one model issues the same kernel sequence); per forward: time inside kernels, time
between a kernel's end and the next kernel's start (launch boundary: cache write-back + dispatch), and both per kernel
name.  No HIP events are involved: these are the dispatch timestamps the profiler reads from the queue."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    """'void anyloc::(anonymous namespace)::gemm_h3_kernel<2, 4, ...>(anyloc::H3Problem, int, int)' -> 'gemm_h3_kernel<2,4,...>'
    (the template arguments tell the four block GEMMs apart: tile shape, ring depth, epilogue)."""
    name = name.replace("(anonymous namespace)::", "").replace("anyloc::", "")
    name = re.sub(r"^void ", "", name.strip())
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).replace(", ", ",").strip()


def main():
    path = sys.argv[1]
    forwards = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a forward's period starts at the patch gather (im2col) and ends where the next one starts: whole periods only (the
    # launches that follow the last im2col of the trace are not a complete period)
    starts = [i for i, r in enumerate(rows) if "im2col" in r[2]]
    if len(starts) < forwards + 1:
        raise SystemExit(f"only {len(starts)} forwards in the trace")
    per = starts[-1] - starts[-2]
    first = starts[-1] - forwards * per
    assert first in starts, "the forwards before the last one do not all issue the same number of launches"
    body = rows[first:starts[-1]]
    period_us = (rows[starts[-1]][0] - rows[first][0]) / 1e3 / forwards
    inside = defaultdict(float)
    after = defaultdict(float)
    calls = defaultdict(int)
    tot_in = tot_gap = host = 0.0
    for f in range(forwards):
        seg = body[f * per:(f + 1) * per]
        nxt = rows[first + (f + 1) * per][0]                      # start of the next period
        gaps = [max(0.0, ((seg[j + 1][0] if j + 1 < per else nxt) - seg[j][1]) / 1e3) for j in range(per)]
        turn = max(range(per), key=lambda j: gaps[j])              # the one long gap of a period: synchronise + host turn-around
        host += gaps[turn]
        for j, (s, e, nme) in enumerate(seg):
            k = short(nme)
            inside[k] += (e - s) / 1e3
            calls[k] += 1
            tot_in += (e - s) / 1e3
            if j != turn:
                after[k] += gaps[j]
                tot_gap += gaps[j]
    span = tot_in + tot_gap
    print(f"# one-image forward (ViT-g/14, 322 x 322, L31 value): dispatch timestamps of {forwards} forwards, {per} launches each\n")
    print(f"start of one forward -> start of the next: {period_us / 1e3:.3f} ms = inside kernels **{tot_in / forwards / 1e3:.3f} ms** "
          f"({100 * tot_in / (period_us * forwards):.1f} %) + between dependent kernels **{tot_gap / forwards / 1e3:.3f} ms** "
          f"({100 * tot_gap / (period_us * forwards):.1f} %; {tot_gap / forwards / (per - 1):.2f} us per launch boundary) + the one long gap "
          f"per forward (synchronise, host turn-around, first launch) {host / forwards / 1e3:.3f} ms; device busy span {span / forwards / 1e3:.3f} ms\n")
    print("| kernel | launches / forward | us inside / launch | us idle after / launch | ms / forward (inside + after) |")
    print("|---|---|---|---|---|")
    for k in sorted(inside, key=lambda k: -(inside[k] + after[k])):
        c = calls[k]
        print(f"| `{k}` | {c // forwards} | {inside[k] / c:.2f} | {after[k] / c:.2f} | {(inside[k] + after[k]) / forwards / 1e3:.3f} |")


if __name__ == "__main__":
    main()
