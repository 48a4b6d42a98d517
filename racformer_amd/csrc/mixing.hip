// mixing.hip -- AdaptiveMixing core as one kernel on the matrix cores (gfx950).
//
// Replaces, per decoder layer, two batched GEMMs, two LayerNorms over [P,64] / [128,64] and two
// ReLUs of AdaptiveMixing.inner_forward (models/racformer_transformer.py:589-603) -- six launches
// that stream the 88 MB sampled features, the 236 MB generated parameters and two 88/118 MB
// intermediates through HBM several times.  Here each (query, group) item is read once
// (x: 24 KB, M: 16 KB, S: 48 KB) and its [128,64] result written once:
//     Y = relu(LN_{[P,64]}(x @ M))        x [P,64], M [64,64]
//     Z = relu(LN_{[128,64]}(S @ Y))      S [128,P]
// parameter_generator and out_proj stay library GEMMs.
//
// Matrix cores: v_mfma_f32_16x16x4_f32 -- f32 in, f32 accumulate, bit-for-bit an fmaf chain, so the
// result has fp32 GEMM accuracy (no bf16 anywhere).  A workgroup = 4 waves = one item; wave w owns
// output columns 16w..16w+15 of both products (6 + 8 accumulator tiles of 16x16).  Operands are
// staged through LDS with row strides chosen so that every MFMA operand read is bank-conflict
// free (x: 68 floats, M / Y: 80, S: P_pad+4); S is staged in two 64-row halves so that two
// workgroups fit in a CU's 160 KB LDS and one's staging overlaps the other's MFMAs.
#include "rac_common.h"

typedef float mix_f4 __attribute__((ext_vector_type(4)));

#define MIX_C 64        /* channels per group (in and out) */
#define MIX_OUT 128     /* out_points */
#define MIX_PMAX 96     /* max in_points (f8: 4 points x 8 frames x 3 depths) */
#define MIX_XS 68       /* sX row stride  */
#define MIX_MS 80       /* sM / sY row stride */
#define MIX_SS (MIX_PMAX + 4) /* sS row stride */
#define MIX_REGION_A (MIX_PMAX * MIX_XS + MIX_C * MIX_MS)  /* sX | sM, later sY, later the output tile */
#define MIX_LDS_FLOATS (MIX_REGION_A + 64 * MIX_SS + 16)

struct MixArgs {
    const float *x;       // [items_q, G, P, 64]
    const float *params;  // row q at params + q*ld_params: per group [64*64 | 128*P]
    float *out;           // [items_q, G, 128, 64]
    int nq, G, P, ld_params;
    float eps;
};

__device__ __forceinline__ float mix_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sum of one value per thread (4 waves); red[] is 4 floats of LDS
__device__ __forceinline__ float mix_block_sum(float v, float *red, int wave, int lane)
{
    v = mix_wave_sum(v);
    __syncthreads();
    if (lane == 0)
        red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256, 2) void mixing_c64_kernel(const MixArgs a)
{
    extern __shared__ float smem[];
    float *sX = smem;                          // [P_pad][68]
    float *sM = smem + MIX_PMAX * MIX_XS;      // [64][80]
    float *sY = smem;                          // [P_pad][80]   (aliases sX|sM after step 1)
    float *sS = smem + MIX_REGION_A;           // [64][P_pad+4]
    float *red = sS + 64 * MIX_SS;             // 4 floats (+pad)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int P = a.P;
    const int MT = (P + 15) >> 4;   // 16-row tiles of x / Y
    const int PP = MIX_PMAX;        // rows / K are always padded to 96 with zeros: branch-free MFMA loops
    (void)MT;
    const int item = blockIdx.x;
    const int q = item / a.G, g = item % a.G;
    const float *gx = a.x + ((size_t)q * a.G + g) * P * MIX_C;
    const float *gM = a.params + (size_t)q * a.ld_params + (size_t)g * (MIX_C * MIX_C + MIX_OUT * P);
    const float *gS = gM + MIX_C * MIX_C;

    // ---- stage x (zero rows up to PP), M and S half 0 -------------------------------------------
    // All global loads of the prologue are issued back to back into registers (16 x 16 B per thread)
    // and only then written to LDS: one memory round trip per workgroup instead of one per loop trip.
    const int ncol4 = PP >> 2;                 // float4 columns of a staged S row
    const bool s_vec = (P & 3) == 0;           // S rows are 16-byte aligned
    auto load_S = [&](int half, int k) -> rac_f4 {
        // element k of this thread's share of S rows 64*half .. +63 (columns zero-padded to PP)
        const int i = tid + 256 * k;
        rac_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < 64 * ncol4) {
            const int r = i / ncol4, c4 = i - r * ncol4;
            const float *src = gS + (size_t)(64 * half + r) * P + c4 * 4;
            if (c4 * 4 + 3 < P) {
                if (s_vec) {
                    v = rac_ld4(src);
                } else {
                    v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                }
            } else {
                if (c4 * 4 + 0 < P) v.x = src[0];
                if (c4 * 4 + 1 < P) v.y = src[1];
                if (c4 * 4 + 2 < P) v.z = src[2];
            }
        }
        return v;
    };
    auto store_S = [&](int k, rac_f4 v) {
        const int i = tid + 256 * k;
        if (i < 64 * ncol4) {
            const int r = i / ncol4, c4 = i - r * ncol4;
            *reinterpret_cast<rac_f4 *>(sS + r * MIX_SS + c4 * 4) = v;
        }
    };
    {
        rac_f4 vx[6], vm[4], vs[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {          // x: up to 96 rows x 16 float4
            const int i = tid + 256 * k, r = i >> 4, c4 = i & 15;
            vx[k] = (rac_f4){0.f, 0.f, 0.f, 0.f};
            if (r < P)
                vx[k] = rac_ld4(gx + r * MIX_C + c4 * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {          // M: 64 rows x 16 float4
            const int i = tid + 256 * k, r = i >> 4, c4 = i & 15;
            vm[k] = rac_ld4(gM + r * MIX_C + c4 * 4);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
            vs[k] = load_S(0, k);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = tid + 256 * k, r = i >> 4, c4 = i & 15;
            *reinterpret_cast<rac_f4 *>(sX + r * MIX_XS + c4 * 4) = vx[k];   // rows >= P are zeros
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k, r = i >> 4, c4 = i & 15;
            *reinterpret_cast<rac_f4 *>(sM + r * MIX_MS + c4 * 4) = vm[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
            store_S(k, vs[k]);
    }
    __syncthreads();

    // ---- step 1: Y = x @ M, wave w -> columns 16w.. ---------------------------------------------
    mix_f4 acc1[6];
#pragma unroll
    for (int m = 0; m < 6; ++m)
        acc1[m] = (mix_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < MIX_C / 4; ++s) {
        const float bv = sM[(4 * s + lk) * MIX_MS + 16 * wave + li];
        float av[6];
#pragma unroll
        for (int m = 0; m < 6; ++m)
            av[m] = sX[(16 * m + li) * MIX_XS + 4 * s + lk];
#pragma unroll
        for (int m = 0; m < 6; ++m)
            acc1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv, acc1[m], 0, 0, 0);
    }
    // LayerNorm over the P x 64 valid elements (rows >= P are padding: exact zeros, excluded)
    float part = 0.f;
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            part += (16 * m + lk * 4 + r < P) ? acc1[m][r] : 0.f;
    const float n1 = (float)(P * MIX_C);
    const float mean1 = mix_block_sum(part, red, wave, lane) / n1;
    part = 0.f;
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = acc1[m][r] - mean1;
            part += (16 * m + lk * 4 + r < P) ? d * d : 0.f;
        }
    const float rstd1 = 1.f / sqrtf(mix_block_sum(part, red, wave, lane) / n1 + a.eps);
    // (the two block sums above end with barriers: every wave is past its last sX / sM read)
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * m + lk * 4 + r;
            const float y = fmaxf((acc1[m][r] - mean1) * rstd1, 0.f);
            sY[row * MIX_MS + 16 * wave + li] = row < P ? y : 0.f;
        }
    __syncthreads();

    // ---- step 2: Z = S @ Y in two 64-row halves --------------------------------------------------
    mix_f4 acc2[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
        acc2[m] = (mix_f4){0.f, 0.f, 0.f, 0.f};
    rac_f4 vs1[6];  // S half 1, fetched while half 0 is being multiplied
#pragma unroll
    for (int k = 0; k < 6; ++k)
        vs1[k] = load_S(1, k);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half == 1) {
            __syncthreads();  // all waves done reading S half 0
#pragma unroll
            for (int k = 0; k < 6; ++k)
                store_S(k, vs1[k]);
            __syncthreads();
        }
#pragma unroll 4
        for (int s = 0; s < MIX_PMAX / 4; ++s) {
            const float bv = sY[(4 * s + lk) * MIX_MS + 16 * wave + li];
            float av[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                av[m] = sS[(16 * m + li) * MIX_SS + 4 * s + lk];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc2[4 * half + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv, acc2[4 * half + m], 0, 0, 0);
        }
    }
    part = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            part += acc2[m][r];
    const float n2 = (float)(MIX_OUT * MIX_C);
    const float mean2 = mix_block_sum(part, red, wave, lane) / n2;
    part = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = acc2[m][r] - mean2;
            part += d * d;
        }
    const float rstd2 = 1.f / sqrtf(mix_block_sum(part, red, wave, lane) / n2 + a.eps);
    // stage the normalised [128][64] tile through LDS (region A is free: Y is dead) for 16-byte stores
    float *sO = smem;  // [128][64], 8192 floats <= MIX_REGION_A
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            sO[(16 * m + lk * 4 + r) * MIX_C + 16 * wave + li] = fmaxf((acc2[m][r] - mean2) * rstd2, 0.f);
    __syncthreads();
    float *go = a.out + ((size_t)q * a.G + g) * MIX_OUT * MIX_C;
    for (int i = tid; i < MIX_OUT * MIX_C / 4; i += 256)
        *reinterpret_cast<rac_f4 *>(go + i * 4) = *reinterpret_cast<const rac_f4 *>(sO + i * 4);
}

extern "C" int rac_mixing_fwd(const float *x, const float *params, float *out, int ld_params, int num_query,
                              int groups, int in_points, int channels, int out_points, float eps, void *stream)
{
    RAC_CHECK_ARG(channels == MIX_C && out_points == MIX_OUT,
                  "rac_mixing_fwd: built for 64 channels per group and 128 out points (got %d, %d)", channels, out_points);
    RAC_CHECK_ARG(in_points >= 1 && in_points <= MIX_PMAX, "rac_mixing_fwd: in_points=%d out of [1,%d]", in_points, MIX_PMAX);
    RAC_CHECK_ARG(num_query >= 0 && groups >= 1, "rac_mixing_fwd: bad sizes");
    RAC_CHECK_ARG(ld_params >= groups * (MIX_C * MIX_C + MIX_OUT * in_points) && ld_params % 4 == 0 &&
                      (MIX_C * MIX_C + MIX_OUT * in_points) % 4 == 0,
                  "rac_mixing_fwd: parameter row stride %d", ld_params);
    static_assert(MIX_REGION_A >= MIX_OUT * MIX_C, "output tile must fit region A");
    static_assert(MIX_REGION_A >= MIX_PMAX * MIX_MS, "Y must fit region A");
    if (num_query == 0)
        return 0;
    RAC_CHECK_ARG(x && params && out, "rac_mixing_fwd: null pointer");
    MixArgs a;
    a.x = x; a.params = params; a.out = out;
    a.nq = num_query; a.G = groups; a.P = in_points; a.ld_params = ld_params; a.eps = eps;
    const size_t lds = (size_t)MIX_LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mixing_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(mixing_c64_kernel, dim3(num_query * groups), dim3(256), lds, (hipStream_t)stream, a);
    return rac_launch_status("rac_mixing_fwd");
}
