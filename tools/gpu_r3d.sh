#!/bin/bash
# round 3, GPU call D: few-query kernel with two slabs in flight (tests + A/B), rocprofv3 kernel stats of the bench command,
# PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) over the ViT-g forward at the bench batch
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$GRAFT_REPO_ROOT
ANYLOC_OPTIONS=topk_fewq_x6=1 timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_distributed_one_gpu.py tests/test_gpu_round3.py -m gpu -q -k "topk or search or sharded" < /dev/null > $O/d_pytest_fewq.log 2>&1; echo "exit: $?" >> $O/d_pytest_fewq.log; tail -5 $O/d_pytest_fewq.log | cut -c1-240
for opt in "topk_fewq_x6=0" "topk_fewq_x6=1" "topk_fewq_x6=0" "topk_fewq_x6=1"; do
  ANYLOC_OPTIONS=$opt timeout 300 python tools/time_topk.py 2>&1 | grep nq | head -1 | sed "s/^/$opt  /"
done | tee $O/d_fewq_ab.log
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes --no-stages < /dev/null > $R/gpurun_out/d_prof_bench.json 2> $R/gpurun_out/d_prof_bench.err
cd $R; f=$(find gpurun_out/d_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/d_kernel_stats.csv && head -12 gpurun_out/d_kernel_stats.csv | cut -c1-220
python tools/bench_brief.py gpurun_out/d_prof_bench.json profiled | head -1 | cut -c1-400
bash tools/gpu_pmc_vit.sh
