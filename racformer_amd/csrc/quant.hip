// quant.hip -- 16-bit block storage of a BEV value stream (gfx950): int16 mantissas with one power-of-two scale per
// (pixel, head) block of 64 channels.
//
// The BEV kernel (bev_fused.hip) is bound by the bytes its taps pull through the CU's texture path, so halving
// the bytes per tap is the lever that remains.  Plain 16-bit floats do not keep the parity budget (bf16 fails on every rig,
// f16 on the random rig's chaotic seed: tests/test_lowprec_storage_gpu.py); a block format does: with the scale shared by the
// 64 channels of one head at one pixel -- exactly what one 16-lane group of the BEV kernel reads per tap -- every value keeps
// 14-15 significant bits relative to the block's largest, and the scale folds into the tap weight (one multiply per tap,
// computed once by the thread that builds the tap list).  value = q * scale, scale = 2^(e-14), e = floor(log2(max |block|)).
// Opt-in (RaCFormerTransformerDecoderLayer.value_storage = "i16"); the default keeps fp32 value streams.
#include "rac_common.h"

__global__ __launch_bounds__(256) void quant_i16_block64_kernel(const float *__restrict__ v, short *__restrict__ q, float *__restrict__ scale,
                                                                long blocks)
{
    // a 16-lane group per block of 64 values (4 per lane); 16 blocks per workgroup and pass
    const int lane16 = threadIdx.x & 15;
    const long stride = (long)gridDim.x * 16;
    for (long b = (long)blockIdx.x * 16 + (threadIdx.x >> 4); b < blocks; b += stride) {
        const rac_f4 x = rac_ld4(v + b * 64 + lane16 * 4);
        unsigned m = rac_absbits4(x.x, x.y, x.z, x.w);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1)
            m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        float up, dn;
        rac_q16_factors(m, up, dn);
        const uint2 o = rac_q16x4(x.x, x.y, x.z, x.w, up);
        *reinterpret_cast<uint2 *>(q + b * 64 + lane16 * 4) = o;
        if (lane16 == 0)
            scale[b] = dn;
    }
}

extern "C" int rac_quant_i16_fwd(const float *values, void *q, float *scale, int64_t blocks, void *stream)
{
    RAC_CHECK_ARG(blocks >= 0, "rac_quant_i16_fwd: blocks=%ld", (long)blocks);
    if (blocks == 0)
        return 0;
    RAC_CHECK_ARG(values && q && scale, "rac_quant_i16_fwd: null pointer");
    RAC_CHECK_ARG(((reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(q)) & 15) == 0, "rac_quant_i16_fwd: pointers must be 16-byte aligned");
    long nb = (blocks + 15) / 16;
    nb = nb > 256 * 32 ? 256 * 32 : nb;
    hipLaunchKernelGGL(quant_i16_block64_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, values,
                       reinterpret_cast<short *>(q), scale, blocks);
    return rac_launch_status("rac_quant_i16_fwd");
}
