#!/bin/bash
# round 6, call H: LayerNorm as the lead role of its consumer GEMM's launch (one image per call): the new tests, then the
# interleaved A/B probe
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "layernorm_lead" > gpurun_out/r6h_pytest_ln_lead.log 2>&1
tail -15 gpurun_out/r6h_pytest_ln_lead.log
timeout 900 python tools/probe_b1_ln_lead.py > gpurun_out/r6h_b1_ln_lead.log 2>&1
cat gpurun_out/r6h_b1_ln_lead.log | tail -40
