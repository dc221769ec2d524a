#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -k "swiglu or scheduling" < /dev/null > $O/g_pytest.log 2>&1; echo "exit: $?" >> $O/g_pytest.log; tail -8 $O/g_pytest.log | cut -c1-240
ANYLOC_OPTIONS=h3_swiglu_t=1 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_vit.py tests/test_gpu_x6.py -m gpu -q < /dev/null > $O/g_pytest_t.log 2>&1; echo "exit: $?" >> $O/g_pytest_t.log; tail -6 $O/g_pytest_t.log | cut -c1-240
REPS=2 bash tools/gpu_ab.sh "h3_swiglu_t=0" "h3_swiglu_t=1"
