#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "kmeans or vlad" 2>&1 | tail -2
cat > /tmp/km.py <<'PY'
import json, os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from anyloc_amd import ops, synth
dev = "cuda"
for (rows, D, K) in ((5_000_000, 1536, 32),):
    x = torch.nn.functional.normalize(torch.randn(rows, D, device=dev))
    c = x[torch.randperm(rows, device=dev)[:K]].clone()
    ops.kmeans_step(x, c, "cosine", True); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): ops.kmeans_step(x, c, "cosine", True)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(json.dumps(dict(v=os.environ.get("ANYLOC_KMEANS_FUSED_V", "2"), rows=rows, D=D, K=K, ms=round(ms, 3), tb_s=round(rows * D * 4 / 1e9 / ms, 3))), flush=True)
PY
for v in 2 1; do ANYLOC_KMEANS_FUSED_V=$v timeout 300 python /tmp/km.py 2>&1 | grep "^{"; done
ANYLOC_VLAD_FUSED=1 timeout 300 python tools/sweep_vlad.py 100000 2>&1 | grep "^{" | head -5
