#!/bin/bash
# round 2: job-pool frames (ZSTD_c_nbWorkers semantics) — parity tests and timing of one large frame
mkdir -p gpurun_out/r02
timeout 120 python -m pytest tests/test_gpu_frames.py -x -q -m gpu > gpurun_out/r02/pytest_frames_mt.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_frames_mt.log
tail -3 gpurun_out/r02/pytest_frames_mt.log
LEVEL=3 SIZE=$((1<<30)) KINDS=datagen JOBS=524288 timeout 100 python scripts/frames_mt_timing.py > gpurun_out/r02/frames_mt_timing_L3.log 2>&1
ZHIP_FRAME_NO_HBM_KERNEL=1 LEVEL=3 SIZE=$((1<<30)) KINDS=datagen JOBS=524288 timeout 100 python scripts/frames_mt_timing.py > gpurun_out/r02/frames_mt_timing_L3_occ2.log 2>&1
LEVEL=1 SIZE=$((1<<30)) KINDS=datagen,text JOBS=0 timeout 100 python scripts/frames_mt_timing.py > gpurun_out/r02/frames_mt_timing.log 2>&1
cat gpurun_out/r02/frames_mt_timing_L3.log gpurun_out/r02/frames_mt_timing_L3_occ2.log gpurun_out/r02/frames_mt_timing.log | cut -c1-260
