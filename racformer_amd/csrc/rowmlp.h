// rowmlp.h -- building blocks for row-wise MLP chains on the matrix cores (gfx950, wave64).
//
// A workgroup (4 waves) owns a tile of 16 query rows and runs a whole chain of small dense layers
// (256..768 -> 10..2189 features) on it without leaving the CU: activations live in LDS, weights
// stream from L2 exactly once per workgroup, each dense layer is a sequence of
// v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate = an fmaf chain, i.e. fp32 GEMM accuracy).
//   A operand  = activations  sIn[row = lane&15][k]        (LDS, ld = K+4)
//   B operand  = weights      W[col = 16*tile + (lane&15)][k]   (global, torch's native [out][in] layout)
//   C/D        = out[row = 4*(lane>>4) + r][col = 16*tile + (lane&15)],  r = 0..3
// Wave w computes column tiles w, w+4, w+8, ...  (out features padded to a multiple of 16 rows of W).
#pragma once
#include "rac_common.h"

typedef float rm_f4 __attribute__((ext_vector_type(4)));

#define RM_ROWS 16
#define RM_LD(K) ((K) + 4) /* LDS row stride of a [16][K] activation tile: 16-byte aligned, low-conflict */

// acc[j] += sIn[16 x K] @ W^T for the column tiles of this wave;  TPW = tiles per wave (compile time).
// W is in torch's native Linear layout [out][in] (ldW = in features).  The k index of an MFMA step is
// only a summation index, so it is assigned such that one 16-byte load feeds four steps: in the group of
// steps 4u..4u+3, lane (li, lk) supplies k = 16u + 4*lk + i for step i -- i.e. the float4
// W[col][16u+4lk .. +3] (global) and sIn[row][16u+4lk .. +3] (LDS), A and B permuted identically.
template <int K, int TPW>
__device__ __forceinline__ void rm_gemm(const float *__restrict__ sIn, int ldIn, const float *__restrict__ W, int ldW,
                                        int wave, int lane, rm_f4 (&acc)[TPW])
{
    const int li = lane & 15, lk = lane >> 4;
    const float *ap = sIn + li * ldIn + 4 * lk;
    const float *bp = W + (size_t)(16 * wave + li) * ldW + 4 * lk;
    constexpr int U = K / 16;
    // Every workgroup streams the same weights: start each one at a different k-block so that the 57
    // tiles do not hit the same L2 lines in lock step (the k order is only a summation order).
    const int u0 = (blockIdx.x * 5) % U;
#pragma unroll 4
    for (int i = 0; i < U; ++i) {
        int u = u0 + i;
        u = u >= U ? u - U : u;
        const rac_f4 a4 = *reinterpret_cast<const rac_f4 *>(ap + 16 * u);
        rac_f4 b4[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            b4[j] = rac_ld4(bp + (size_t)(64 * j) * ldW + 16 * u);
        // k-major order: consecutive MFMAs hit different accumulators (40-cycle dependent latency)
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4[j].x, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4[j].y, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4[j].z, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4[j].w, acc[j], 0, 0, 0);
    }
}

// out[row][col] = act(acc + bias[col] (+ res[row][col]));  act: 0 none, 1 relu
template <int TPW>
__device__ __forceinline__ void rm_store(const rm_f4 (&acc)[TPW], const float *__restrict__ bias, const float *sRes, int ldRes,
                                         float *sOut, int ldOut, int relu, int wave, int lane)
{
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int col = 16 * (wave + 4 * j) + li;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * lk + r;
            float v = acc[j][r] + bv;
            if (sRes)
                v += sRes[row * ldRes + col];
            if (relu)
                v = fmaxf(v, 0.f);
            sOut[row * ldOut + col] = v;
        }
    }
}

__device__ __forceinline__ float rm_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// In-place LayerNorm(256) (+ReLU) of the 16 rows of an LDS tile: wave w normalises rows 4w..4w+3.
__device__ __forceinline__ void rm_layernorm256(float *sT, int ld, const float *__restrict__ gamma,
                                                const float *__restrict__ beta, float eps, int relu, int wave, int lane)
{
    const rac_f4 g = rac_ld4(gamma + lane * 4), b = rac_ld4(beta + lane * 4);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        float *row = sT + (4 * wave + rr) * ld + lane * 4;
        rac_f4 x = *reinterpret_cast<const rac_f4 *>(row);
        const float mean = rm_wave_sum((x.x + x.y) + (x.z + x.w)) / 256.f;
        const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
        const float rstd = 1.f / sqrtf(rm_wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 256.f + eps);
        rac_f4 y;
        y.x = d0 * rstd * g.x + b.x;
        y.y = d1 * rstd * g.y + b.y;
        y.z = d2 * rstd * g.z + b.z;
        y.w = d3 * rstd * g.w + b.w;
        if (relu) {
            y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
        }
        *reinterpret_cast<rac_f4 *>(row) = y;
    }
}

// Load a [16][256] tile of a row-major [n][256] global array into LDS (rows >= n: zeros).
__device__ __forceinline__ void rm_load_tile256(const float *__restrict__ g, int row0, int n, float *sT, int ld, int tid)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k, r = i >> 6, c4 = i & 63;
        rac_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < n)
            v = rac_ld4(g + (size_t)(row0 + r) * 256 + c4 * 4);
        *reinterpret_cast<rac_f4 *>(sT + r * ld + c4 * 4) = v;
    }
}

__device__ __forceinline__ void rm_store_tile256(float *__restrict__ g, int row0, int n, const float *sT, int ld, int tid)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k, r = i >> 6, c4 = i & 63;
        if (row0 + r < n)
            *reinterpret_cast<rac_f4 *>(g + (size_t)(row0 + r) * 256 + c4 * 4) = *reinterpret_cast<const rac_f4 *>(sT + r * ld + c4 * 4);
    }
}
