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
#include "s4d_device.h"
#include <string.h>

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
    float *out;           // [items_q, G, 128, 64] (may be null when out_split is given)
    _Float16 *out_split;  // optional: f16 [items_q, G*256, hi 32 | lo 32] = the line image of out * split_scale (rac_outproj_fwd's A operand)
    int nq, G, P, ld_params;
    float eps, split_scale;
    float param_scale;    // every generated parameter is multiplied by this on load (the split GEMM's power-of-two alpha)
};

// rac_mixing_sampled_fwd: the mixing kernel gathers its own x -- the adaptive 4D sampling of the item's T * P points (s4d_device.h)
// runs inside the workgroup while the item's 64 KB of generated parameters are on their way from HBM
struct MixSampArgs {
    MixArgs m;      // (x unused when s.feat[0] != nullptr)
    S4dArgs s;      // as for rac_sampling4d_fwd; out unused; P = points per frame, m.P = T * P
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

// copy the finished [128][64] tile from LDS to the fp32 output and / or the f16 split image
__device__ __forceinline__ void mix_write_out(const MixArgs &a, const float *sO, int q, int g, int tid)
{
    if (a.out) {
        float *go = a.out + ((size_t)q * a.G + g) * MIX_OUT * MIX_C;
        for (int i = tid; i < MIX_OUT * MIX_C / 4; i += 256)
            *reinterpret_cast<rac_f4 *>(go + i * 4) = *reinterpret_cast<const rac_f4 *>(sO + i * 4);
    }
    if (a.out_split) {
        // A operand of out_proj as a 3-product split GEMM on the f16 matrix cores (rac_outproj_fwd): per 32 values of K one
        // 128-byte line [hi 32 | lo 32] of out * split_scale.  K = (g, out point, channel), so the item's 128 out points
        // are 256 consecutive lines: 32 KB contiguous per item, every value stored once.
        _Float16 *go = a.out_split + ((size_t)q * a.G + g) * (size_t)(MIX_OUT * 2 * MIX_C);
        for (int i = tid; i < MIX_OUT * MIX_C / 4; i += 256) {
            const rac_f4 v = *reinterpret_cast<const rac_f4 *>(sO + i * 4);
            rac_h4 hi, lo;
            rac_split_f16(v.x * a.split_scale, hi.x, lo.x);
            rac_split_f16(v.y * a.split_scale, hi.y, lo.y);
            rac_split_f16(v.z * a.split_scale, hi.z, lo.z);
            rac_split_f16(v.w * a.split_scale, hi.w, lo.w);
            const int o = i >> 4, c = (i & 15) * 4;
            _Float16 *dst = go + (o * 2 + (c >> 5)) * 64 + (c & 31);
            *reinterpret_cast<rac_h4 *>(dst) = hi;
            *reinterpret_cast<rac_h4 *>(dst + 32) = lo;
        }
    }
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
        v.x *= a.param_scale; v.y *= a.param_scale; v.z *= a.param_scale; v.w *= a.param_scale;
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
            vm[k].x *= a.param_scale; vm[k].y *= a.param_scale; vm[k].z *= a.param_scale; vm[k].w *= a.param_scale;
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
    mix_write_out(a, sO, q, g, tid);
}

// ------------------------------------------------------------------------------------------------
// Split-precision variant: the same two products on the 16-bit matrix cores (v_mfma_f32_16x16x32_{bf16,f16}) with
// every operand split into 16-bit terms and the leading cross products accumulated in fp32 -- fp32-GEMM accuracy
// at a fraction of the matrix-core time of the f32-input MFMA, which leaves the kernel bound by its 0.4 GB of
// HBM traffic per launch.
//   * x @ M: both operands as three bf16 terms (24 significant bits, fp32 exponent range: the sampled image
//     features are data, nothing bounds them), the six products down to 2^-16 relative kept (truncation 2^-23);
//   * S @ Y: two f16 terms each (22 bits), products hi*hi + hi*lo + lo*hi (truncation 2^-22).  Y is LayerNorm
//     output (|Y| < 79); |S| is bounded by the caller from the generator's weights (AdaptiveMixing.split_packs);
//   * x and S are converted once while being staged (LDS holds the 16-bit images, conflict-free row strides
//     of 160 / 224 bytes for ds_read_b128 fragment reads);
//   * wave w's slab of M (64 x 16) goes straight from global memory into B fragments (no LDS);
//   * Y never leaves the registers: a 16x16 accumulator tile has its column on the lane and rows 4*lk..4*lk+3 in
//     its 4 registers, so tiles (2t, 2t+1) ARE the B fragment of k-step t of the second product once converted,
//     with k = 32t + 16h + 4lk + i  <->  fragment element 4h + i.  S is staged with that same k permutation
//     (an 8-byte-granular shuffle inside each 32-wide block), so its A fragments stay single 16-byte reads.
typedef _Float16 mix_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 mix_b8 __attribute__((ext_vector_type(8)));
struct alignas(8) mix_b4 {
    __bf16 x, y, z, w;
};
// v = t1 + t2 + t3 exactly to 24 bits, each term a bf16 (round-to-nearest)
__device__ __forceinline__ void mix_split_bf16(float v, __bf16 &t1, __bf16 &t2, __bf16 &t3)
{
    t1 = (__bf16)v;
    const float r1 = v - (float)t1;
    t2 = (__bf16)r1;
    t3 = (__bf16)(r1 - (float)t2);
}

#ifndef RAC_MIX_REVERSE
#define RAC_MIX_REVERSE 1   /* 1: walk the (query, group) items from the LAST one: the generator wrote those rows last, so part of them is still in the
                                Infinity Cache (A/B, profiles/r04_mixing_reverse_ab.json: 96.4-97.7 -> 91.2-92.2 us per launch); 0: in write order */
#endif
#define MIXH_XS 80    /* f16 row stride of the x images  (160 B) */
#define MIXH_SS 104   /* f16 row stride of the S images  (208 B) */
#define MIXH_X_BYTES (3 * MIX_PMAX * MIXH_XS * 2)   /* three bf16 terms: 46080 */
#define MIXH_S_BYTES (2 * MIX_OUT * MIXH_SS * 2)    /* hi + lo of all 128 rows: 53248 */
/* the x images (step 1), the S images (step 2) and the output tile (epilogue) take turns in ONE region: 53 KB, so three
   workgroups share a CU's 160 KB */
#define MIXH_REGION_BYTES (MIXH_S_BYTES > MIXH_X_BYTES ? MIXH_S_BYTES : MIXH_X_BYTES)
#define MIXH_LDS_BYTES (MIXH_REGION_BYTES + 64)

__device__ __forceinline__ void mix_split4(const rac_f4 v, float scale, rac_h4 &hi, rac_h4 &lo)
{
    rac_split_f16(v.x * scale, hi.x, lo.x);
    rac_split_f16(v.y * scale, hi.y, lo.y);
    rac_split_f16(v.z * scale, hi.z, lo.z);
    rac_split_f16(v.w * scale, hi.w, lo.w);
}

#ifndef MSG_LB
#define MSG_LB 2   /* levels per load batch of the in-kernel gather (8 taps in flight per lane): the register budget of three workgroups per CU */
#endif
// LS = 0: x [items, P, 64] is read from memory (rac_mixing_fwd).  LS = L > 0: x is gathered here from the L-level feature pyramid
// (rac_mixing_sampled_fwd): after issuing the loads of its parameters the workgroup computes the item's T * P keypoints and tap
// tables (one thread each, s4d_keypoint / s4d_taps_of_level: the code of the stand-alone sampling kernel), gathers them with the
// lane mapping the x staging below already has (a 16-lane group per point, 4 channels per lane, 6 points per group) -- the same
// taps, weights and FMA order as sampling4d_c64_kernel, so x is bit-identical to what that kernel would have written -- and
// converts the sums straight into the bf16 operand images.  The 88 MB x tensor of a layer never exists, and the gather (bound
// by the CU's texture path) runs under the parameter stream (bound by HBM) of the other workgroups of the CU.
template <int LS>
__global__ __launch_bounds__(256, 3) void mixing_c64_f16x3_kernel(const MixSampArgs A)
{
    const MixArgs &a = A.m;
    extern __shared__ float smem[];
    unsigned char *lds = reinterpret_cast<unsigned char *>(smem);
    __bf16 *sX1 = reinterpret_cast<__bf16 *>(lds);                         // [96][80] x 3 terms
    __bf16 *sX2 = sX1 + MIX_PMAX * MIXH_XS;
    __bf16 *sX3 = sX2 + MIX_PMAX * MIXH_XS;
    _Float16 *sSh = reinterpret_cast<_Float16 *>(lds);                     // [128][104] hi, then lo: over the x images
    _Float16 *sSl = sSh + MIX_OUT * MIXH_SS;
    float *red = reinterpret_cast<float *>(lds + MIXH_REGION_BYTES);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int P = a.P;
    const int item = RAC_MIX_REVERSE ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int q = item / a.G, g = item % a.G;
    const float *gx = a.x + ((size_t)q * a.G + g) * P * MIX_C;
    const float *gM = a.params + (size_t)q * a.ld_params + (size_t)g * (MIX_C * MIX_C + MIX_OUT * P);
    const float *gS = gM + MIX_C * MIX_C;
    const float ps = a.param_scale;

    // ---- all global loads of the item are issued up front (22 x 16 B + 16 x 4 B per thread) ------------------
    const bool s_vec = (P & 3) == 0;
    auto load_S = [&](int half, int k) -> rac_f4 {     // float4 number tid+256k of S rows 64*half.. (24 per row, zero-padded)
        const int i = tid + 256 * k;
        const int r = i / 24, c4 = i - r * 24;
        const float *src = gS + (size_t)(64 * half + r) * P + c4 * 4;
        rac_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (c4 * 4 + 3 < P) {
            if (s_vec) {
                v = rac_ld4_stream(src);
            } else {
                v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
            }
        } else {
            if (c4 * 4 + 0 < P) v.x = src[0];
            if (c4 * 4 + 1 < P) v.y = src[1];
            if (c4 * 4 + 2 < P) v.z = src[2];
        }
        return v;
    };
    auto store_S = [&](_Float16 *dh, _Float16 *dl, int k, const rac_f4 v) {
        const int i = tid + 256 * k;
        const int r = i / 24, c4 = i - r * 24;
        // k = 4*c4 = 32t + 16h + 4lk'  ->  column 32t + 8lk' + 4h  (the accumulator-as-operand k order)
        const int t = c4 >> 3, rem = c4 & 7, col = 32 * t + 8 * (rem & 3) + 4 * (rem >> 2);
        rac_h4 hi, lo;
        mix_split4(v, ps, hi, lo);
        *reinterpret_cast<rac_h4 *>(dh + r * MIXH_SS + col) = hi;
        *reinterpret_cast<rac_h4 *>(dl + r * MIXH_SS + col) = lo;
    };
    rac_f4 vx[6], vs0[6], vs1[6];
    float mv[2][8];
    if (LS == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = tid + 256 * k, r = i >> 4, c4 = i & 15;
            vx[k] = (rac_f4){0.f, 0.f, 0.f, 0.f};
            if (r < P)
                vx[k] = rac_ld4_stream(gx + r * MIX_C + c4 * 4);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            mv[ks][j] = RAC_STREAM_NT ? __builtin_nontemporal_load(gM + (32 * ks + 8 * lk + j) * MIX_C + 16 * wave + li) : gM[(32 * ks + 8 * lk + j) * MIX_C + 16 * wave + li];
#pragma unroll
    for (int k = 0; k < 6; ++k)
        vs0[k] = load_S(0, k);
    if (LS == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            vs1[k] = load_S(1, k);
    }
    if (LS > 0) {
        // ---- x gathered in place (the parameter loads above are in flight meanwhile) ---------------------------------------
        constexpr int L = LS > 0 ? LS : 1;
        const S4dArgs &sa = A.s;
        const int Ppt = sa.P, T = sa.T, N = sa.N;
        const int b = q / sa.Q, qq = q - b * sa.Q;
        // LDS for this phase, over the region the x images take afterwards: tap table [L][96][4 offsets | 4 weights], the
        // lidar2img matrices of the T frames, one flag per point (some tap inside a map)
        float *stab = smem;
        constexpr int lstride = MIX_PMAX * 8;
        float *sl2i = stab + L * lstride;                                           // [T*N][16]
        unsigned char *sval = reinterpret_cast<unsigned char *>(sl2i + T * N * 16); // [96]
        for (int i = tid; i < T * N * 16; i += 256)
            sl2i[i] = sa.l2i[(size_t)b * T * N * 16 + i];
        __syncthreads();
        for (int i = tid; i < P; i += 256) {
            const int t = i / Ppt, p = i - t * Ppt;
            float loc3[3], wl[L];
            s4d_keypoint<L>(sa, sl2i + t * N * 16, b, t, g, qq, p, loc3, wl);
            const float lu = loc3[0], lv = loc3[1];
            const int view = (int)loc3[2] & 255;
            const unsigned slot_in_b = (unsigned)(t * sa.G + g);                   // the item's slot of frame t inside batch element b
            bool any = false;
#pragma unroll
            for (int l = 0; l < L; ++l)
                any = s4d_taps_of_level<float>(sa.H[l], sa.W[l], lu, lv, view, wl[l], slot_in_b * sa.feat_bytes[l], stab + l * lstride + i * 8) || any;
            sval[i] = any ? 1 : 0;
            if (sa.loc_out) {
                const size_t sl = ((size_t)b * T + t) * sa.G + g;
                float *lo = sa.loc_out + ((sl * sa.Q + qq) * Ppt + p) * 3;
                lo[0] = lu;
                lo[1] = lv;
                lo[2] = (float)((int)loc3[2] >> 8) / (float)max(N - 1, 1);
                float *wo = sa.w_out + ((sl * sa.Q + qq) * Ppt + p) * L;
#pragma unroll
                for (int l = 0; l < L; ++l)
                    wo[l] = wl[l];
            }
        }
        __syncthreads();
        // one descriptor per level over the T * G slots of batch element b (tap offsets carry the slot; the host checked that
        // they stay below S4D_TAP_OUTSIDE); b through readfirstlane: a scalar descriptor, no waterfall loop around the loads
        const int b_uni = __builtin_amdgcn_readfirstlane(b);
        __amdgpu_buffer_rsrc_t rsrc[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const unsigned bytes_b = (unsigned)(T * sa.G) * sa.feat_bytes[l];
            rsrc[l] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(sa.feat[l]) + (size_t)b_uni * bytes_b), 0,
                                                        bytes_b, 0x00020000);
        }
        const unsigned lane_off = (unsigned)((tid & 15) * 16);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int r = (tid >> 4) + 16 * k;            // the row (point) this 16-lane group stages as vx[k]
            rac_acc4 acc4 = rac_acc4_zero();
            if (r < P && sval[r]) {
                const float *e = stab + r * 8;
#pragma unroll
                for (int l0 = 0; l0 < L; l0 += MSG_LB) {
                    rac_f4 v[MSG_LB][4], tw[MSG_LB];
#pragma unroll
                    for (int u = 0; u < MSG_LB; ++u) {
                        const int l = l0 + u;
                        if (l < L) {
                            const s4d_u4 o = *reinterpret_cast<const s4d_u4 *>(e + l * lstride);
                            tw[u] = *reinterpret_cast<const rac_f4 *>(e + l * lstride + 4);
                            v[u][0] = s4d_tap<float>(rsrc[l], o.x + lane_off);
                            v[u][1] = s4d_tap<float>(rsrc[l], o.y + lane_off);
                            v[u][2] = s4d_tap<float>(rsrc[l], o.z + lane_off);
                            v[u][3] = s4d_tap<float>(rsrc[l], o.w + lane_off);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < MSG_LB; ++u) {
                        if (l0 + u < L) {
                            const float w4[4] = {tw[u].x, tw[u].y, tw[u].z, tw[u].w};
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                rac_tap_fma(acc4, v[u][c].x, v[u][c].y, v[u][c].z, v[u][c].w, w4[c]);
                        }
                    }
                }
            }
            rac_acc4_get(acc4, vx[k].x, vx[k].y, vx[k].z, vx[k].w);
        }
        __syncthreads();      // every wave is past its last tap-table read: the region takes the x images now
        // (the second half of S is requested only now: its 24 registers were the gather's; it lands under step 1)
#pragma unroll
        for (int k = 0; k < 6; ++k)
            vs1[k] = load_S(1, k);
    }

#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int i = tid + 256 * k, r = i >> 4, c4 = i & 15;
        mix_b4 t1, t2, t3;                                                    // rows >= P are zeros
        mix_split_bf16(vx[k].x, t1.x, t2.x, t3.x);
        mix_split_bf16(vx[k].y, t1.y, t2.y, t3.y);
        mix_split_bf16(vx[k].z, t1.z, t2.z, t3.z);
        mix_split_bf16(vx[k].w, t1.w, t2.w, t3.w);
        *reinterpret_cast<mix_b4 *>(sX1 + r * MIXH_XS + c4 * 4) = t1;
        *reinterpret_cast<mix_b4 *>(sX2 + r * MIXH_XS + c4 * 4) = t2;
        *reinterpret_cast<mix_b4 *>(sX3 + r * MIXH_XS + c4 * 4) = t3;
    }
    mix_b8 bM1[2], bM2[2], bM3[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 t1, t2, t3;
            mix_split_bf16(mv[ks][j] * ps, t1, t2, t3);
            bM1[ks][j] = t1;
            bM2[ks][j] = t2;
            bM3[ks][j] = t3;
        }
    __syncthreads();

    // ---- step 1: Y = x @ M, wave w -> columns 16w.. ------------------------------------------------------------
    mix_f4 acc1[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        acc1[m] = (mix_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = (16 * m + li) * MIXH_XS + 32 * ks + 8 * lk;
            const mix_b8 a1 = *reinterpret_cast<const mix_b8 *>(sX1 + off);
            const mix_b8 a2 = *reinterpret_cast<const mix_b8 *>(sX2 + off);
            const mix_b8 a3 = *reinterpret_cast<const mix_b8 *>(sX3 + off);
            // smallest terms first
            acc1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, bM1[ks], acc1[m], 0, 0, 0);
            acc1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bM2[ks], acc1[m], 0, 0, 0);
            acc1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bM3[ks], acc1[m], 0, 0, 0);
            acc1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bM1[ks], acc1[m], 0, 0, 0);
            acc1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bM2[ks], acc1[m], 0, 0, 0);
            acc1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bM1[ks], acc1[m], 0, 0, 0);
        }
    }
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
    // (both block sums end with barriers: every wave is past its last read of the x images, so the region can
    //  take the S images now; S stayed in registers through step 1)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        store_S(sSh, sSl, k, vs0[k]);
        store_S(sSh + 64 * MIXH_SS, sSl + 64 * MIXH_SS, k, vs1[k]);
    }
    // Y = relu(LN(.)) -> B fragments of the second product, in registers
    mix_h8 bYh[3], bYl[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 2 * t + h, row = 16 * m + lk * 4 + r;
                const float y = row < P ? fmaxf((acc1[m][r] - mean1) * rstd1, 0.f) : 0.f;
                _Float16 yh, yl;
                rac_split_f16(y, yh, yl);
                bYh[t][4 * h + r] = yh;
                bYl[t][4 * h + r] = yl;
            }
    __syncthreads();   // S visible

    // ---- step 2: Z = S @ Y ---------------------------------------------------------------------------------------
    mix_f4 acc2[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        acc2[m] = (mix_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int off = (16 * m + li) * MIXH_SS + 32 * t + 8 * lk;
            const mix_h8 ah = *reinterpret_cast<const mix_h8 *>(sSh + off);
            const mix_h8 al = *reinterpret_cast<const mix_h8 *>(sSl + off);
            acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bYh[t], acc2[m], 0, 0, 0);
            acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bYl[t], acc2[m], 0, 0, 0);
            acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bYh[t], acc2[m], 0, 0, 0);
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
    // every wave is past its S reads: stage the normalised [128][64] tile (32 KB) over the x / S images
    float *sO = smem;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            sO[(16 * m + lk * 4 + r) * MIX_C + 16 * wave + li] = fmaxf((acc2[m][r] - mean2) * rstd2, 0.f);
    __syncthreads();
    mix_write_out(a, sO, q, g, tid);
}

extern "C" int rac_mixing_fwd(const float *x, const float *params, float param_scale, float *out, void *out_split,
                              float split_scale, int ld_params, int num_query, int groups, int in_points, int channels, int out_points,
                              float eps, int mfma_mode, void *stream)
{
    RAC_CHECK_ARG(mfma_mode == RAC_MIX_F32 || mfma_mode == RAC_MIX_F16X3, "rac_mixing_fwd: mfma_mode=%d", mfma_mode);
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
    RAC_CHECK_ARG(x && params && (out || out_split), "rac_mixing_fwd: null pointer");
    MixArgs a;
    a.x = x; a.params = params; a.out = out;
    a.out_split = reinterpret_cast<_Float16 *>(out_split); a.split_scale = split_scale; a.param_scale = param_scale;
    a.nq = num_query; a.G = groups; a.P = in_points; a.ld_params = ld_params; a.eps = eps;
    const size_t lds = (size_t)MIX_LDS_FLOATS * sizeof(float);
    if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_MIXING_F32, reinterpret_cast<const void *>(mixing_c64_kernel), (int)((int)lds)))
        return rc_attr;
    if (mfma_mode == RAC_MIX_F16X3) {
        static_assert(3 * MIXH_LDS_BYTES <= 160 * 1024, "three workgroups per CU");
        static_assert(MIXH_REGION_BYTES >= MIX_OUT * MIX_C * 4, "output tile must fit the shared region");
        static_assert(MIXH_SS >= MIX_PMAX + 8 && (MIXH_SS * 2) % 16 == 0, "S row stride");
        if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_MIXING_F16, reinterpret_cast<const void *>(mixing_c64_f16x3_kernel<0>), (int)((int)MIXH_LDS_BYTES)))
            return rc_attr;
        MixSampArgs A;
        memset(&A.s, 0, sizeof(A.s));
        A.m = a;
        hipLaunchKernelGGL(mixing_c64_f16x3_kernel<0>, dim3(num_query * groups), dim3(256), MIXH_LDS_BYTES, (hipStream_t)stream, A);
        return rac_launch_status("rac_mixing_fwd");
    }
    hipLaunchKernelGGL(mixing_c64_kernel, dim3(num_query * groups), dim3(256), lds, (hipStream_t)stream, a);
    return rac_launch_status("rac_mixing_fwd");
}

// AdaptiveMixing core with the adaptive 4D sampling inside (models/racformer_transformer.py:361-408 + sparsebev_sampling.py:45-131 +
// msmv_sampling_forward.cu:75-164 feeding :589-603): arguments = those of rac_sampling4d_fwd (without `out`: the sampled features
// stay on chip) followed by those of rac_mixing_fwd (without `x`).  Built for what the decoder runs: fp32 pyramid of 4 levels,
// 64 channels per group, T * NP * D <= 96 points per item, split-precision MFMA mode.
extern "C" int rac_mixing_sampled_fwd(const void *const *feats, const int32_t *hw, int L, const float *query_bbox, const float *box_table,
                                      const float *offsets, const float *ray_logits, const float *scale_logits, const float *time_diff,
                                      const float *lidar2img, float *loc_out, float *w_out, const unsigned char *view_in, int ld_off,
                                      int ld_ray, int ld_scale, int B, int T, int N, int G, int Q, int NP, int D, int C,
                                      const float *pc_range, const float *depth_base, float d_region, float image_h, float image_w,
                                      float eps_proj, int dtype, const float *params, float param_scale, float *out, void *out_split,
                                      float split_scale, int ld_params, int out_points, float eps_ln, void *stream)
{
    RAC_CHECK_ARG(L == 4, "rac_mixing_sampled_fwd: L=%d (built for 4 levels)", L);
    RAC_CHECK_ARG(C == MIX_C && out_points == MIX_OUT, "rac_mixing_sampled_fwd: built for 64 channels per group and 128 out points (got %d, %d)", C, out_points);
    RAC_CHECK_ARG(dtype == RAC_F32, "rac_mixing_sampled_fwd: fp32 feature maps only (dtype %d)", dtype);
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && T >= 1 && N >= 1 && N <= S4D_MAX_CAMS && G >= 1 && NP >= 1 && D >= 1 && D <= S4D_MAX_DEPTH,
                  "rac_mixing_sampled_fwd: bad sizes B=%d T=%d N=%d G=%d Q=%d NP=%d D=%d", B, T, N, G, Q, NP, D);
    const int Ppt = NP * D, P = T * Ppt;
    RAC_CHECK_ARG(P <= MIX_PMAX, "rac_mixing_sampled_fwd: T * NP * D = %d points per item (at most %d)", P, MIX_PMAX);
    RAC_CHECK_ARG(ld_params >= G * (MIX_C * MIX_C + MIX_OUT * P) && ld_params % 4 == 0 && (MIX_C * MIX_C + MIX_OUT * P) % 4 == 0,
                  "rac_mixing_sampled_fwd: parameter row stride %d", ld_params);
    RAC_CHECK_ARG(ld_off >= G * Ppt * 3 && ld_ray >= D && ld_scale >= G * T * Ppt * L, "rac_mixing_sampled_fwd: row strides too small");
    RAC_CHECK_ARG((loc_out == nullptr) == (w_out == nullptr), "rac_mixing_sampled_fwd: loc_out and w_out go together");
    if (B == 0 || Q == 0)
        return 0;
    RAC_CHECK_ARG(feats && hw && query_bbox && box_table && offsets && ray_logits && scale_logits && time_diff && lidar2img && pc_range &&
                      depth_base && params && (out || out_split), "rac_mixing_sampled_fwd: null pointer");
    RAC_CHECK_ARG((size_t)L * MIX_PMAX * 32 + (size_t)T * N * 64 + MIX_PMAX <= (size_t)MIXH_REGION_BYTES,
                  "rac_mixing_sampled_fwd: T * N = %d camera matrices do not fit beside the tap table", T * N);
    MixSampArgs A;
    memset(&A, 0, sizeof(A));
    S4dArgs &s = A.s;
    for (int l = 0; l < RAC_MAX_LEVELS; ++l) {
        s.H[l] = s.W[l] = 1;
    }
    for (int l = 0; l < L; ++l) {
        RAC_CHECK_ARG(feats[l] != nullptr && hw[2 * l] >= 1 && hw[2 * l + 1] >= 1, "rac_mixing_sampled_fwd: level %d", l);
        s.feat[l] = feats[l];
        s.H[l] = hw[2 * l];
        s.W[l] = hw[2 * l + 1];
        const size_t bytes = (size_t)N * s.H[l] * s.W[l] * 64 * 4;      // one slot's maps
        RAC_CHECK_ARG(bytes * T * G < (size_t)S4D_TAP_OUTSIDE,
                      "rac_mixing_sampled_fwd: the T * G slots of level %d hold %zu bytes per sample (the tap offsets are 31-bit)", l, bytes * T * G);
        s.feat_bytes[l] = (unsigned)bytes;
    }
    s.qbox = query_bbox; s.box = box_table; s.off = offsets; s.ray = ray_logits; s.scale = scale_logits;
    s.time_diff = time_diff; s.l2i = lidar2img; s.out = nullptr; s.loc_out = loc_out; s.w_out = w_out; s.view_in = view_in;
    for (int i = 0; i < S4D_MAX_DEPTH; ++i)
        s.depth_base[i] = i < D ? depth_base[i] : 0.f;
    for (int i = 0; i < 6; ++i)
        s.pc[i] = pc_range[i];
    s.d_region = d_region; s.image_h = image_h; s.image_w = image_w; s.eps = eps_proj;
    s.L = L; s.B = B; s.T = T; s.N = N; s.G = G; s.Q = Q; s.NP = NP; s.D = D; s.P = Ppt;
    s.ld_off = ld_off; s.ld_ray = ld_ray; s.ld_scale = ld_scale;
    MixArgs &a = A.m;
    a.x = nullptr; a.params = params; a.out = out;
    a.out_split = reinterpret_cast<_Float16 *>(out_split); a.split_scale = split_scale; a.param_scale = param_scale;
    a.nq = B * Q; a.G = G; a.P = P; a.ld_params = ld_params; a.eps = eps_ln;
    if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_MIXING_SAMPLED, reinterpret_cast<const void *>(mixing_c64_f16x3_kernel<4>), (int)MIXH_LDS_BYTES))
        return rc_attr;
    hipLaunchKernelGGL(mixing_c64_f16x3_kernel<4>, dim3(B * Q * G), dim3(256), MIXH_LDS_BYTES, (hipStream_t)stream, A);
    return rac_launch_status("rac_mixing_sampled_fwd");
}
