#!/bin/bash
# fused3 (fp16 screening + exact resolution) against the exact-score kernels: k-means step, VLAD sweep, phase stamps
mkdir -p gpurun_out
{
echo "## k-means step, 5 M x 1536, K=32 (tools/time_kmeans.py): default = fused3, 2 = kmeans_fused2_kernel, 1 = vlad_fused_kernel"
for v in 0 2 1; do for d in "" clustered; do ANYLOC_KMEANS_FUSED_V=$v timeout 300 python tools/time_kmeans.py $d 2>&1 | grep '"rows"'; done; done
echo "## other shapes, fused3"
timeout 300 python tools/time_kmeans.py all clustered 2>&1 | grep '"rows"' | tail -2
echo "## hard VLAD, 529 x 1536 tokens per image, K=32 (tools/sweep_vlad.py): default, then ANYLOC_VLAD_FUSED_V=1"
timeout 300 python tools/sweep_vlad.py 100000 2>&1 | grep '"vlad"'
ANYLOC_VLAD_FUSED_V=1 timeout 300 python tools/sweep_vlad.py 100000 2>&1 | grep '"vlad"' | sed 's/^/V=1 /'
echo "## phase stamps of fused3, unit 0 (tools/stamp_kmeans.py), isotropic then clustered rows"
for d in "" clustered; do timeout 200 python tools/stamp_kmeans.py 2000000 $d 2>&1 | tail -8 | cut -c1-140; done
} | tee gpurun_out/r2_fused3_ab.log
