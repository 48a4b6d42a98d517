// regroup.hip -- FPN pyramid regroup to the sampling layout, for gfx950 (MI355X).
//
// Replaces the reshape/permute/.contiguous() of RaCFormerTransformerDecoder.forward
// (models/racformer_transformer.py:112-124, channel-last branch):
//     in  [B, T*N, G*C, H, W]  ->  out [B*T*G, N, H, W, C]
// i.e. for every (b,t,n,g) a [C][H*W] -> [H*W][C] transpose.  HBM-bound (reads and writes the
// whole 735 MB pyramid once per forward): 64x64 tiles through LDS (row stride 65 dwords, so the
// transposed read is bank-conflict free), 256-byte coalesced segments on both sides.
#include "rac_common.h"

template <typename OT>
__device__ __forceinline__ void regroup_store(OT *p, float v);
template <>
__device__ __forceinline__ void regroup_store<float>(float *p, float v) { *p = v; }
template <>
__device__ __forceinline__ void regroup_store<unsigned short>(unsigned short *p, float v)
{
    // round-to-nearest-even f32 -> bf16 (NaN stays NaN via the quiet bit)
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) {
        *p = (unsigned short)((u >> 16) | 0x40);
        return;
    }
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}

template <typename OT>
__global__ __launch_bounds__(256) void regroup_kernel(const float *__restrict__ in, OT *__restrict__ out,
                                                      int T, int N, int G, int C, int HW)
{
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const int hw0 = blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    int z = blockIdx.z;  // ((b*T+t)*N+n)*G+g
    const int g = z % G; z /= G;
    const int n = z % N; z /= N;
    const int t = z % T;
    const int b = z / T;
    const float *src = in + ((((size_t)b * T + t) * N + n) * G + g) * (size_t)C * HW;
    OT *dst = out + ((((size_t)b * T + t) * G + g) * N + n) * (size_t)HW * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = c0 + ty * 16 + r, hw = hw0 + tx;
        if (c < C && hw < HW)
            tile[ty * 16 + r][tx] = src[(size_t)c * HW + hw];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int hw = hw0 + ty * 16 + r, c = c0 + tx;
        if (c < C && hw < HW)
            regroup_store<OT>(dst + (size_t)hw * C + c, tile[tx][ty * 16 + r]);
    }
}

extern "C" int rac_regroup_fwd(const float *in, void *out, int B, int T, int N, int G, int C, int H,
                               int W, int out_dtype, void *stream)
{
    RAC_CHECK_ARG(in && out, "rac_regroup_fwd: null pointer");
    RAC_CHECK_ARG(B >= 0 && T >= 1 && N >= 1 && G >= 1 && C >= 1 && H >= 1 && W >= 1,
                  "rac_regroup_fwd: bad sizes");
    RAC_CHECK_ARG(out_dtype == RAC_F32 || out_dtype == RAC_BF16, "rac_regroup_fwd: dtype %d", out_dtype);
    const long nz = (long)B * T * N * G;
    RAC_CHECK_ARG(nz <= 65535, "rac_regroup_fwd: B*T*N*G=%ld exceeds grid.z", nz);
    if (nz == 0)
        return 0;
    const int HW = H * W;
    dim3 grid((HW + 63) / 64, (C + 63) / 64, (unsigned)nz);
    if (out_dtype == RAC_F32)
        hipLaunchKernelGGL(regroup_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, in, (float *)out, T, N, G, C, HW);
    else
        hipLaunchKernelGGL(regroup_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, in,
                           (unsigned short *)out, T, N, G, C, HW);
    return rac_launch_status("rac_regroup_fwd");
}
