cd /root/repo
scripts/gpu_r4_full.sh
scripts/pmc_sq.sh r04_L1_datagen 1 1024 > gpurun_out/pmc_sq_final.log 2>&1
tail -3 gpurun_out/pmc_sq_final.log
scripts/pmc_legs.sh > gpurun_out/pmc_legs_final.log 2>&1
tail -3 gpurun_out/pmc_legs_final.log
