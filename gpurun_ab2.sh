cd /root/repo
run() { ZHIP_LIB=$PWD/zstd_amd/variants/$1.so ZHIP_FAST_GWAVES_DENSE=$2 AB_SET=1 timeout 120 python scripts/ab_dense.py 1024 text,silesia 2>&1 | grep -v amdgpu.ids; }
run q4g5 7; run q4g5 8; run q4g5 9; run q4g6 9; run q4g6 11; run q5g5 9; run q5g5 11
