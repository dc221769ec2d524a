#!/bin/bash
export TMPDIR=/tmp
for opt in "h3_mfma16=0" "h3_mfma16=1" "h3_mfma16=0" "h3_mfma16=1"; do
  ANYLOC_OPTIONS=$opt timeout 300 python tools/run_stage.py config3_shard --check 2>&1 | grep -E "config3|rror" | cut -c1-400
done | tee gpurun_out/h3m_config3.log
