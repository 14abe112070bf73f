// zhip_k_decode.hip — translation unit of the decode kernels (zhip_kernels_decode.h); device code only, launched from zhip_lib.hip
#include "zhip_kernels_decode.h"
