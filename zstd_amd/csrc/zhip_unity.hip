// zhip_unity.hip — the whole library as ONE translation unit: profiling builds (-DZHIP_PROF: the phase counters are one __device__ array)
// and A/B variants (scripts/build_variant.sh).  The product build (zstd_amd/build.py) compiles zhip_lib.hip and each zhip_k_*.hip separately.
#define ZHIP_UNITY 1
#include "zhip_lib.hip"
