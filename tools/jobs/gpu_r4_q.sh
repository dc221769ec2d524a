#!/bin/bash
# round 4, GPU call Q: is the fabric traffic of the block GEMMs joules that matter?  The XCD scheduling group (option
# h3_group_m: tile-rows whose workgroups are co-resident on one XCD) changes how often A / W panels are re-fetched; per
# setting: FETCH_SIZE of the real B = 61 forward (rocprofv3 --pmc, own pass) and the time per step of the plain bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/gm_write -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/gm_write.log 2>&1
for gm in 2 4 8 16 32; do
  cd /tmp
  ANYLOC_OPTIONS="h3_group_m=$gm" timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/gm_fetch_$gm -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/gm_fetch_$gm.log 2>&1
  cd $R
  rm -f gpurun_out/gm_traffic_$gm.json
  python tools/pmc_traffic.py gpurun_out/gm_fetch_$gm gpurun_out/gm_write h3 gpurun_out/gm_traffic_$gm.json > /dev/null 2>&1
  ANYLOC_OPTIONS="h3_group_m=$gm" timeout 300 python bench.py --steps 8 --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > gpurun_out/gm_bench_$gm.json 2>> gpurun_out/gm_bench.err
  python - <<PY
import json
t = json.load(open("gpurun_out/gm_traffic_$gm.json"))["h3"]
b = json.loads([l for l in open("gpurun_out/gm_bench_$gm.json") if l.startswith("{")][-1])
k = b["roofline"]["kernels_ms_per_step"]
print("h3_group_m=$gm: " + "  ".join(f"{n[4:-5]} {t[n]['bytes_per_launch'] / 1e9:.2f} GB ({t[n]['refetch_ratio']:.2f}x) {k[n] / 32:.3f} ms" for n in ("vit_w12_gemm", "vit_qkv_gemm", "vit_fc2_gemm", "vit_proj_gemm") if n in t)
      + f"  | {b['value']:.1f} images/s", flush=True)
PY
done
rm -rf gpurun_out/gm_fetch_* gpurun_out/gm_write
