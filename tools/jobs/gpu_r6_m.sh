#!/bin/bash
# round 6, call M: screened retrieval in the bench stages (config3_shard with the oracle check, config3_whole_db) and the config3 workload
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_screen.py -x -q > gpurun_out/r6m_pytest_screen.log 2>&1
tail -3 gpurun_out/r6m_pytest_screen.log
timeout 1200 python tools/run_stage.py config3_shard --check > gpurun_out/r6m_stage_config3_shard.json 2> gpurun_out/r6m_stage_config3_shard.err
tail -c 3000 gpurun_out/r6m_stage_config3_shard.json; tail -3 gpurun_out/r6m_stage_config3_shard.err
timeout 1500 python tools/run_stage.py config3_whole_db > gpurun_out/r6m_stage_config3_whole_db.json 2> gpurun_out/r6m_stage_config3_whole_db.err
tail -c 1500 gpurun_out/r6m_stage_config3_whole_db.json; tail -3 gpurun_out/r6m_stage_config3_whole_db.err
timeout 1200 python bench.py --workload config3 --steps 3 --warmup 1 > gpurun_out/r6m_bench_config3.json 2> gpurun_out/r6m_bench_config3.err
tail -c 2500 gpurun_out/r6m_bench_config3.json; tail -3 gpurun_out/r6m_bench_config3.err
