// zhip_k_parse.hip — translation unit of the parse kernels (zhip_kernels_parse.h); device code only, launched from zhip_lib.hip
#include "zhip_kernels_parse.h"
