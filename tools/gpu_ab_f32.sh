export TMPDIR=/tmp; mkdir -p gpurun_out
cp anyloc_amd/libanyloc_hip.so /tmp/lib_orig.so
for rep in 1 2; do for v in base f32chain; do
  cp tools/ab_libs/lib_$v.so anyloc_amd/libanyloc_hip.so
  timeout 300 python bench.py --gemm f32 --steps 4 --warmup 1 --no-modes --no-stages --no-cpu-baseline < /dev/null > gpurun_out/abf_${v}_$rep.json 2>> gpurun_out/abf.err
  python tools/bench_brief.py gpurun_out/abf_${v}_$rep.json "f32mode:$v#$rep" | head -1 | cut -c1-260
done; done
cp /tmp/lib_orig.so anyloc_amd/libanyloc_hip.so
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -k "driver" 2>&1 | tail -2
