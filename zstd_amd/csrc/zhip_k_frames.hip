// zhip_k_frames.hip — translation unit of the frames kernels (zhip_kernels_frames.h); device code only, launched from zhip_lib.hip
#include "zhip_kernels_frames.h"
