// zhip_kernels.h — every __global__ entry point in one include: the host SIMT emulator (tests/simt) and the single-object builds
// (zhip_unity.hip: profiling and A/B variants) use this; the product library compiles each family as its own translation unit (zhip_k_*.hip).
#pragma once
#include "zhip_kernels_parse.h"
#include "zhip_kernels_lazy.h"
#include "zhip_kernels_entropy.h"
#include "zhip_kernels_frames.h"
#include "zhip_kernels_decode.h"
