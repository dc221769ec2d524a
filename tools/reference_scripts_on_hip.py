"""Run the reference's own driver scripts UNMODIFIED on the HIP path (one GPU), each in a fresh interpreter through
``python -m anyloc_amd.run``, on synthetic datasets / synthetic ViT-S weights:

    ANYLOC_REFERENCE_ROOT=/path/to/AnyLoc python tools/reference_scripts_on_hip.py

The reference tree is not part of this repository and does not exist on the driver's GPU box; the log of a run against
a transient copy is profiles/r03_reference_scripts_on_hip.log.  One JSON line per script (return code, the recall /
shape lines it printed, files it cached) and a final ``ALL OK`` / ``FAILED``."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ANYLOC_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_synth_dataset  # noqa: E402


def run(script, args, want, tmp):
    env = dict(os.environ, ANYLOC_SYNTHETIC_WEIGHTS="0", PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-m", "anyloc_amd.run", os.path.join(REF, "scripts", script)] + args,
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = res.stdout
    ok = res.returncode == 0 and "Traceback" not in out and "Unhandled exception" not in out and all(w in out for w in want)
    lines = [ln.strip() for ln in out.splitlines() if re.search(r"R@\d|shape: torch.Size|Recall", ln)][:12]
    pts = sorted({f for dp, _, fs in os.walk(tmp) for f in fs if f.endswith(".pt")})[:8]
    print(json.dumps(dict(script=script, ok=ok, returncode=res.returncode, lines=lines, cached=pts)), flush=True)
    if not ok:
        print(out[-3000:], res.stderr[-2000:], flush=True)
    return ok


def main():
    import torch
    assert torch.cuda.is_available(), "needs the GPU (this is the HIP-path run)"
    from anyloc_amd import _lib
    _lib.load()
    print(json.dumps(dict(device=torch.cuda.get_device_name(0), library=os.path.relpath(_lib.LIB_PATH, ROOT), reference_root=os.path.basename(REF.rstrip("/")))), flush=True)
    good = True
    with tempfile.TemporaryDirectory() as t:
        data = os.path.join(t, "data")
        make_synth_dataset.write(data, "st_lucia", n_db=6, n_qu=3, h=112, w=140)
        common = ["--prog.data-vg-dir", data, "--prog.vg-dataset-name", "st_lucia", "--model-type", "dinov2_vits14"]
        good &= run("dino_v2_vlad.py", common + ["--prog.cache-dir", os.path.join(t, "c1"), "--desc-layer", "9", "--desc-facet", "value",
                                                 "--num-clusters", "4", "--bd-args.resize", "112", "140", "--exp-id", "t1",
                                                 "--top-k-vals", "1", "2", "3", "--cache-vlad-descs"],
                    ["Database VLADs shape: torch.Size([6, 1536])", "Query VLADs shape: torch.Size([3, 1536])"], t)
        make_synth_dataset.write(data, "pitts30k", n_db=4, n_qu=2, h=112, w=140, seed=2)
        good &= run("dino_v2_global_vocab_vlad.py", common + ["--prog.cache-dir", os.path.join(t, "c2"), "--desc-layer", "9",
                                                              "--desc-facet", "value", "--num-clusters", "4", "--vlad-cache-dir",
                                                              os.path.join(t, "vc"), "--db-samples.st-lucia", "1", "--db-samples.pitts30k", "2",
                                                              "--exp-id", "g1", "--top-k-vals", "1", "2"], ["R@1", "END"], t)
        rec = common + ["--prog.cache-dir", os.path.join(t, "c3"), "--bd-args.resize", "112", "140", "--top-k-vals", "1", "2", "3"]
        for method in ("average", "max"):
            good &= run("dino_v2_gp.py", rec + ["--desc-layer", "9", "--desc-facet", "value", "--pool-method", method],
                        ["Generated pooled descriptors", "R@1"], t)
        good &= run("dino_v2_gem.py", rec + ["--desc-layer", "9", "--desc-facet", "value", "--gem-p", "3"],
                    ["Database GeMs shape: torch.Size([6, 384])", "R@1"], t)
        good &= run("dino_v2_global_vpr.py", rec, ["R@1"], t)
    print("ALL OK" if good else "FAILED", flush=True)
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()
