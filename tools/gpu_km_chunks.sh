#!/bin/bash
# k-means step at the config-4 size against the number of chunks (partial sums written / re-read per chunk)
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for c in 2048 1024 512 256; do for d in clustered ""; do echo "chunks<=$c"; ANYLOC_KMEANS_MAX_CHUNKS=$c timeout 120 python tools/time_kmeans.py $d < /dev/null 2>&1 | grep '"rows"'; done; done
ANYLOC_KMEANS_MAX_CHUNKS=2048 timeout 200 python tools/bench_configs.py 4 < /dev/null 2>&1 | grep "^{" | cut -c1-400
ANYLOC_KMEANS_MAX_CHUNKS=512 timeout 200 python tools/bench_configs.py 4 < /dev/null 2>&1 | grep "^{" | cut -c1-400
} | tee gpurun_out/km_chunks.log
