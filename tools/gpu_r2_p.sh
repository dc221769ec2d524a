#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "kmeans" 2>&1 | tail -1
timeout 300 python tools/stamp_kmeans.py 2>&1 | tail -7
timeout 300 python tools/time_kmeans.py 2>&1 | grep "^{"
