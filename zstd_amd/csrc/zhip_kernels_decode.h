// zhip_kernels_decode.h — __global__ entry points: the decoder (k_decode: one frame per workgroup; k_bf_*: one large frame, block-parallel).
// Compiled into its own code object by zhip_k_decode.hip: a change in another kernel family cannot move this one's inlining or register allocation
// (round 3 ended on a decoder whose code the block-parallel decoder's arrival had reshaped).  Declarations for the host side: zhip_kernel_decls.h.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_kernel_params.h"
#include "zhip_decode.h"
#include "zhip_decode_big.h"

namespace zhip {

// Decoder: persistent 128-thread workgroups, each takes frames from a queue (counter) until it is empty; per workgroup a
// literal buffer and two hand-over buffers of sequence records in HBM/L2.  Dynamic LDS = sizeof(DecShared).
#ifndef ZHIP_DEC_WAVES_PER_EU
#define ZHIP_DEC_WAVES_PER_EU 3          /* 6 workgroups per CU = what the 25 KB of LDS per workgroup allow; keeps the register allocator at <= 168 VGPRs */
#endif
__global__ void __launch_bounds__(ZHIP_DEC_THREADS) __attribute__((amdgpu_waves_per_eu(ZHIP_DEC_WAVES_PER_EU, ZHIP_DEC_WAVES_PER_EU)))
k_decode(const uint8_t* __restrict__ src, const ZhipDFrame* __restrict__ frames, uint32_t nFrames, uint8_t* __restrict__ dst,
         uint8_t* __restrict__ litArena, ZhipDSeq* __restrict__ recArena, uint32_t* __restrict__ counter,
         ZhipDDictDev dict, const uint64_t* __restrict__ defTabs, ZhipDResult* __restrict__ results)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    DecShared* const S = (DecShared*)smem;
    uint8_t* const litBuf = litArena + (size_t)blockIdx.x * ZHIP_DEC_LIT_STRIDE;
    ZhipDSeq* const recBuf = recArena + (size_t)blockIdx.x * 2 * (ZHIP_DEC_CHUNK + 1);
    if (threadIdx.x == 0) { S->dictHufIn = 0; S->dictFseIn = 0; }
    for (;;) {
        ZHIP_CONVERGE();
        if (threadIdx.x == 0) S->frame = atomicAdd(counter, 1u);
        __syncthreads();
        uint32_t const f = ZHIP_UNIFORM(S->frame);             // scalar: the loop's exit is one s_cbranch for the whole wavefront (zhip_decode.h, decode_frame)
        __syncthreads();
        if (f >= nFrames) break;
        ZhipDFrame const fr = frames[f];
        decode_frame(S, src + fr.srcOff, fr.srcLen, dst + fr.dstOff, fr.dstCap, litBuf, recBuf, dict, dict.content != nullptr, defTabs, results + f);
    }
}

// ONE large frame, block-parallel (zhip_decode_big.h): src = the frame, out = its content; the launches in order
__global__ void __launch_bounds__(64)
k_bf_walk(const uint8_t* __restrict__ src, uint32_t srcLen, uint32_t hdrSize, uint32_t blockMax, uint32_t hasChecksum,
          ZhipBfBlock* __restrict__ blocks, uint32_t capBlocks, ZhipBfInfo* __restrict__ info)
{
    bf_walk(src, srcLen, hdrSize, blockMax, hasChecksum, blocks, capBlocks, info);
}
__global__ void __launch_bounds__(256)
k_bf_prep(const uint8_t* __restrict__ src, uint32_t blockMax, ZhipBfBlock* __restrict__ blocks, const ZhipBfInfo* __restrict__ info)
{
    uint32_t const bi = blockIdx.x * 256 + threadIdx.x;
    if (bi < info->nBlocks) bf_prep(src, blockMax, blocks, bi);
}
__global__ void __launch_bounds__(64)
k_bf_deps(ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info) { bf_deps(blocks, info); }
// dynamic LDS = sizeof(DecShared); grid = number of blocks
__global__ void __launch_bounds__(ZHIP_BF_THREADS)
k_bf_entropy(const uint8_t* __restrict__ src, uint32_t blockMax, ZhipBfBlock* __restrict__ blocks, const ZhipBfInfo* __restrict__ info,
             uint8_t* __restrict__ litArena, ZhipDSeq* __restrict__ recArena, const uint64_t* __restrict__ defTabs)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    if (blockIdx.x < ZHIP_UNIFORM(info->nBlocks)) bf_entropy_block((DecShared*)smem, src, blockMax, blocks, blockIdx.x, litArena, recArena, defTabs);
}
__global__ void __launch_bounds__(64)
k_bf_scan(ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info, uint32_t dstCap)
{
    __shared__ uint32_t sh[64 * 3];
    bf_scan(blocks, info, dstCap, sh);
}
__global__ void __launch_bounds__(256)
k_bf_build(const uint8_t* __restrict__ src, const ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info, const uint8_t* __restrict__ litArena,
           const ZhipDSeq* __restrict__ recArena, uint8_t* __restrict__ out, uint32_t* __restrict__ map)
{
    if (blockIdx.x < info->nBlocks && info->status == 0) bf_build_block(src, blocks, blockIdx.x, litArena, recArena, out, map, info);
}
__global__ void __launch_bounds__(256)
k_bf_jump(uint32_t* __restrict__ map, uint32_t n, ZhipBfInfo* __restrict__ info) { bf_jump(map, n, &info->changed); }
__global__ void __launch_bounds__(256)
k_bf_copy(const uint32_t* __restrict__ map, uint8_t* __restrict__ out, uint32_t n) { bf_copy(map, out, n); }

// content checksums: checks[] = XXH64 low words of the decoded frames (k_xxh64 over the destination)
__global__ void __launch_bounds__(256)
k_dec_verify(ZhipDResult* __restrict__ results, const uint32_t* __restrict__ checks, uint32_t nFrames)
{
    uint32_t const i = blockIdx.x * 256 + threadIdx.x;
    if (i < nFrames && results[i].status == 0 && results[i].hasChecksum && results[i].checksum != checks[i]) { results[i].status = ZHIP_DE_CHECKSUM; results[i].size = 0; }
}

}  // namespace zhip
