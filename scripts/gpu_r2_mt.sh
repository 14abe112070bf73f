#!/bin/bash
# round 2: job-pool frames (ZSTD_c_nbWorkers semantics) — parity tests and timing of one large frame
mkdir -p gpurun_out/r02
timeout 30 python scripts/jobdbg.py 2>&1 | tail -1 || exit 1
timeout 120 python -m pytest tests/test_gpu_frames.py -x -q -m gpu > gpurun_out/r02/pytest_frames_mt.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_frames_mt.log
tail -3 gpurun_out/r02/pytest_frames_mt.log
LEVEL=1 SIZE=$((1<<30)) KINDS=datagen timeout 100 python scripts/frames_mt_timing.py > gpurun_out/r02/frames_mt_timing.log 2>&1
cat gpurun_out/r02/frames_mt_timing.log | cut -c1-260
