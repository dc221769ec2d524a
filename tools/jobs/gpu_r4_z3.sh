#!/bin/bash
# round 4, last GPU minutes: the 192 x 128 small-M plan of the one-image w12 GEMM -- A/B, then the tests of the small-M plans
# (every tile configuration incl. the new one, forced on all block GEMMs) and the one-image call pattern on the new default.
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 200 python tools/probe_b1_w12_tall.py > gpurun_out/rz3_b1_w12_tall.log 2> gpurun_out/rz3_probe.err; echo "probe exit $?"
cut -c1-240 gpurun_out/rz3_b1_w12_tall.log; tail -3 gpurun_out/rz3_probe.err | cut -c1-300
timeout 200 python -m pytest tests/test_gpu_vit.py -m gpu -q -x -p no:cacheprovider -k "small or plans or batch or b1 or one_image" < /dev/null > gpurun_out/rz3_pytest_small.log 2>&1; echo "pytest(small) exit $?"; tail -4 gpurun_out/rz3_pytest_small.log | cut -c1-300
timeout 150 python -m pytest tests/test_gpu_vit.py tests/test_gpu_round4.py -m gpu -q -x -p no:cacheprovider -k "not small and not plans and not rccl and not eight_ranks and not config2_size and not config3_panel" < /dev/null > gpurun_out/rz3_pytest_rest.log 2>&1; echo "pytest(rest) exit $?"; tail -4 gpurun_out/rz3_pytest_rest.log | cut -c1-300
