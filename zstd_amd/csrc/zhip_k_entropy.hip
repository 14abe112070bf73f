// zhip_k_entropy.hip — translation unit of the entropy kernels (zhip_kernels_entropy.h); device code only, launched from zhip_lib.hip
#include "zhip_kernels_entropy.h"
