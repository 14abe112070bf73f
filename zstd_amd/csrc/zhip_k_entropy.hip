// zhip_k_entropy.hip — translation unit of the entropy kernels (zhip_kernels_entropy.h); device code only, launched from zhip_lib.hip
#include "zhip_kernels_entropy.h"
#ifdef ZHIP_PAD_KB      /* measurement only (scripts/build_variant.sh ... -DZHIP_PAD_KB=n): n KB of s_nop in this code object — does the layout of the library move another family's speed? */
namespace zhip { __global__ void k_pad() { asm volatile(".rept %0\n s_nop 0\n .endr" :: "n"(ZHIP_PAD_KB * 256)); } }
#endif
