#!/bin/bash
# round 5, call M: the reference's own driver scripts, unmodified, on the HIP path of this build (a transient, untracked copy of the
# reference's .py files under _ref_tmp/, removed after the call), and the test the driver's box has to skip
mkdir -p gpurun_out
export ANYLOC_REFERENCE_ROOT=$PWD/_ref_tmp
timeout 900 python tools/reference_scripts_on_hip.py 2>&1 | cut -c1-1200 | tee gpurun_out/r5m_reference_scripts.log
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -k reference_dino_v2_vlad_script < /dev/null 2>&1 | tail -3 | tee -a gpurun_out/r5m_reference_scripts.log
