cd /tmp; export TMPDIR=/tmp
ROOT=/root/repo
mkdir -p $ROOT/gpurun_out/pmc_ent
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -f csv -d $ROOT/gpurun_out/pmc_ent/sq -o sq -- python $ROOT/bench.py --level 1 --mib 512 --workload text --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined-extra --no-extra-legs > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace -f csv -d $ROOT/gpurun_out/pmc_ent/sq2 -o sq2 -- python $ROOT/bench.py --level 1 --mib 512 --workload text --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined-extra --no-extra-legs > /dev/null 2>&1
python $ROOT/scripts/pmc_summary.py $ROOT/gpurun_out/pmc_ent | grep -A18 "k_entropy"
