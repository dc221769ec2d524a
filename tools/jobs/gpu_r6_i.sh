#!/bin/bash
# round 6, call I: LayerNorm as lead workgroups inside the batched GEMM launches: the new tests, then the interleaved A/B probe
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "layernorm_lead" > gpurun_out/r6i_pytest_ln_lead.log 2>&1
tail -15 gpurun_out/r6i_pytest_ln_lead.log
timeout 900 python tools/probe_batched_ln_lead.py > gpurun_out/r6i_batched_ln_lead.log 2>&1
tail -12 gpurun_out/r6i_batched_ln_lead.log
