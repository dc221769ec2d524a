"""Run single `stages` entries of bench.py (same code, same timing) without the headline workload:
    python tools/run_stage.py config3_shard kmeans_5Mx1536 vlad_61img vlad_256img vitl_518_2taps [--check]
Prints one JSON object per stage.  Kernel variants through ANYLOC_OPTIONS (include/anyloc_hip.h)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from anyloc_amd import _lib, synth  # noqa: E402


class _Vocab:
    def __init__(self, dev):
        self.c_centers = 0.7 * synth.clustered_tokens(1, 32, 1536, n_modes=32, seed=3, device=str(dev))[0]


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["vlad_61img", "vlad_256img", "kmeans_5Mx1536", "config3_shard"]
    check = "--check" in sys.argv
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _lib.load()
    for n in names:
        if n.startswith("vlad_"):
            res = bench.stage_vlad(dev, _Vocab(dev), int(n.split("_")[1].replace("img", "")), check)
        elif n.startswith("vitl_518_2taps"):           # vitl_518_2taps[:batch]
            res = bench.stage_vitl(dev, check, *([int(n.split(":")[1])] if ":" in n else []))
        elif n == "config3_whole_db":
            res = bench.stage_config3_whole_db(dev)
        else:
            res = {"kmeans_5Mx1536": bench.stage_kmeans, "config3_shard": bench.stage_config3_shard}[n](dev, check)
        print(json.dumps({n: res, "options": os.environ.get("ANYLOC_OPTIONS", "")}), flush=True)


if __name__ == "__main__":
    main()
