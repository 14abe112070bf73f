// zhip_k_lazy.hip — translation unit of the lazy kernels (zhip_kernels_lazy.h); device code only, launched from zhip_lib.hip
#include "zhip_kernels_lazy.h"
