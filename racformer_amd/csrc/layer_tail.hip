// layer_tail.hip -- everything after the three sampling kernels of a decoder layer, in ONE launch (gfx950).
//
// Per 16-query tile, on one CU, with activations in LDS and exact-fp32 MFMA dense layers (rowmlp.h):
//   radar = LN(x1 + output_proj_r(bev_r))                 models/bev_self_attention.py:221-225, racformer_transformer.py:249
//   lss   = LN(x1 + output_proj_l(bev_l))                 :251
//   x2    = LN2(x1 + sum_s out_proj_partial[s] + b_out)   AdaptiveMixing tail :606-608 + norm2 :256
//   f     = LN_f(fusion([x2 | radar | lss]))              :257
//   x3    = LN3(f + W2 relu(W1 f + b1) + b2)              mmcv FFN(256,512) + norm3 :258
//   cls   = W6 relu(LN(W3 relu(LN(W0 x3))))               cls_branch :199-205, :260
//   delta = V4 relu(V2 relu(V0 x3))                       reg_branch :207-212, :261
//   refine_bbox + velocity / time_diff + theta_d2xy       :230-236, :265-269, :134
// The reference (and the op-decomposed plan) needs ~75 launches for this (13 small GEMMs, 8 LayerNorms,
// adds, ReLUs, cat, the refine elementwise chain); here the 3.5 MB of weights are read once per tile
// straight from L2 and nothing but x3 / cls / boxes returns to HBM.
#include <cstddef>
#include "rowmlp.h"

#define LT_E 256
#define LT_LD RM_LD(LT_E)       /* 260 */
#define LT_LDCAT RM_LD(3 * LT_E) /* 772: [x2 | radar | lss], later the 512-wide FFN hidden */
#define LT_TWO_PI 6.283185307179586f
#define LT_NUM_WEIGHTS 37

struct TailArgs {
    // activations
    const float *x1, *bev_r, *bev_l, *partials, *qbox, *td_safe;
    float *x3, *cls, *pred, *xy;
    // weights in torch's native [out][in] layout (10-wide heads zero-padded to 16 rows), biases, LayerNorm affine pairs
    const float *Wor, *bor, *Wol, *bol, *b_mix;
    const float *g_r, *be_r, *g_l, *be_l, *g_2, *be_2;
    const float *Wf, *bf, *g_f, *be_f;
    const float *W1, *b1, *W2, *b2, *g_3, *be_3;
    const float *Wc0, *bc0, *g_c1, *be_c1, *Wc3, *bc3, *g_c4, *be_c4, *Wc6, *bc6;
    const float *Wr0, *br0, *Wr2, *br2, *Wr4, *br4;
    long pstride;
    int n, S, Q, T, num_classes, code_size;
    float eps, num_ray;
};

__device__ __forceinline__ float lt_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float lt_inverse_sigmoid(float x)
{
    x = fminf(fmaxf(x, 0.f), 1.f);
    return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}

__global__ __launch_bounds__(256) void layer_tail_kernel(const TailArgs a)
{
    extern __shared__ float smem[];
    float *sCat = smem;                       // [16][772]: x2 | radar | lss ; later FFN hidden [16][516]
    float *sA = sCat + RM_ROWS * LT_LDCAT;    // [16][260]: x1, later branch temp 2
    float *sB = sA + RM_ROWS * LT_LD;         // [16][260]: bev tile / branch temp 1
    float *sF = sB + RM_ROWS * LT_LD;         // [16][260]: f
    float *sX = sF + RM_ROWS * LT_LD;         // [16][260]: x3
    float *sD = sX + RM_ROWS * LT_LD;         // [16][20]: 10-wide head outputs
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * RM_ROWS, n = a.n;

    // ---- radar / lss: LN(x1 + bev @ Wo^T + bo) into the concat tile ------------------------------
    rm_load_tile256(a.x1, row0, n, sA, LT_LD, tid);
    rm_load_tile256(a.bev_r, row0, n, sB, LT_LD, tid);
    __syncthreads();
    {
        rm_f4 acc[4] = {};
        rm_gemm<LT_E, 4>(sB, LT_LD, a.Wor, LT_E, wave, lane, acc);
        rm_store<4>(acc, a.bor, sA, LT_LD, sCat + LT_E, LT_LDCAT, 0, wave, lane);
    }
    __syncthreads();
    rm_layernorm256(sCat + LT_E, LT_LDCAT, a.g_r, a.be_r, a.eps, 0, wave, lane);
    rm_load_tile256(a.bev_l, row0, n, sB, LT_LD, tid);
    __syncthreads();
    {
        rm_f4 acc[4] = {};
        rm_gemm<LT_E, 4>(sB, LT_LD, a.Wol, LT_E, wave, lane, acc);
        rm_store<4>(acc, a.bol, sA, LT_LD, sCat + 2 * LT_E, LT_LDCAT, 0, wave, lane);
    }
    // ---- x2 = LN2(x1 + sum_s partial[s] + b_mix): elementwise, thread -> (row = tid/16.., 4 cols) -----
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k, r = i >> 6, c4 = i & 63;
        rac_f4 v = *reinterpret_cast<const rac_f4 *>(sA + r * LT_LD + c4 * 4);
        const rac_f4 bm = rac_ld4(a.b_mix + c4 * 4);
        v.x += bm.x; v.y += bm.y; v.z += bm.z; v.w += bm.w;
        if (row0 + r < n) {
            const float *pp = a.partials + (size_t)(row0 + r) * LT_E + c4 * 4;
            rac_f4 sum = rac_ld4(pp);
            for (int s = 1; s < a.S; ++s) {
                const rac_f4 w = rac_ld4(pp + (size_t)s * a.pstride);
                sum.x += w.x; sum.y += w.y; sum.z += w.z; sum.w += w.w;
            }
            v.x += sum.x; v.y += sum.y; v.z += sum.z; v.w += sum.w;
        }
        *reinterpret_cast<rac_f4 *>(sCat + r * LT_LDCAT + c4 * 4) = v;
    }
    __syncthreads();
    rm_layernorm256(sCat + 2 * LT_E, LT_LDCAT, a.g_l, a.be_l, a.eps, 0, wave, lane);
    rm_layernorm256(sCat, LT_LDCAT, a.g_2, a.be_2, a.eps, 0, wave, lane);
    __syncthreads();

    // ---- f = LN_f(fusion(cat)) ---------------------------------------------------------------------
    {
        rm_f4 acc[4] = {};
        rm_gemm<3 * LT_E, 4>(sCat, LT_LDCAT, a.Wf, 3 * LT_E, wave, lane, acc);
        rm_store<4>(acc, a.bf, nullptr, 0, sF, LT_LD, 0, wave, lane);
    }
    __syncthreads();
    rm_layernorm256(sF, LT_LD, a.g_f, a.be_f, a.eps, 0, wave, lane);
    __syncthreads();

    // ---- FFN: x3 = LN3(f + W2 relu(W1 f + b1) + b2) -------------------------------------------------
    float *sH = sCat;  // [16][516] hidden (the concat tile is dead)
#pragma unroll
    for (int half = 0; half < 2; ++half) {   // 512 outputs as two 256-wide passes
        rm_f4 acc[4] = {};
        rm_gemm<LT_E, 4>(sF, LT_LD, a.W1 + (size_t)half * LT_E * LT_E, LT_E, wave, lane, acc);
        rm_store<4>(acc, a.b1 + half * LT_E, nullptr, 0, sH + half * LT_E, RM_LD(2 * LT_E), 1, wave, lane);
    }
    __syncthreads();
    {
        rm_f4 acc[4] = {};
        rm_gemm<2 * LT_E, 4>(sH, RM_LD(2 * LT_E), a.W2, 2 * LT_E, wave, lane, acc);
        rm_store<4>(acc, a.b2, sF, LT_LD, sX, LT_LD, 0, wave, lane);
    }
    __syncthreads();
    rm_layernorm256(sX, LT_LD, a.g_3, a.be_3, a.eps, 0, wave, lane);
    __syncthreads();
    rm_store_tile256(a.x3, row0, n, sX, LT_LD, tid);

    // ---- cls branch ----------------------------------------------------------------------------------
    {
        rm_f4 acc[4] = {};
        rm_gemm<LT_E, 4>(sX, LT_LD, a.Wc0, LT_E, wave, lane, acc);
        rm_store<4>(acc, a.bc0, nullptr, 0, sB, LT_LD, 0, wave, lane);
    }
    __syncthreads();
    rm_layernorm256(sB, LT_LD, a.g_c1, a.be_c1, a.eps, 1, wave, lane);
    __syncthreads();
    {
        rm_f4 acc[4] = {};
        rm_gemm<LT_E, 4>(sB, LT_LD, a.Wc3, LT_E, wave, lane, acc);
        rm_store<4>(acc, a.bc3, nullptr, 0, sA, LT_LD, 0, wave, lane);
    }
    __syncthreads();
    rm_layernorm256(sA, LT_LD, a.g_c4, a.be_c4, a.eps, 1, wave, lane);
    __syncthreads();
    if (wave == 0) {
        rm_f4 acc[1] = {};
        rm_gemm<LT_E, 1>(sA, LT_LD, a.Wc6, LT_E, 0, lane, acc);
        rm_store<1>(acc, a.bc6, nullptr, 0, sD, 20, 0, 0, lane);
    }
    // ---- reg branch (first layer overlaps with the cls head on waves 1-3 of the next barrier) ---------
    __syncthreads();
    if (tid < RM_ROWS * 16) {
        const int r = tid >> 4, c = tid & 15;
        if (c < a.num_classes && row0 + r < n)
            a.cls[(size_t)(row0 + r) * a.num_classes + c] = sD[r * 20 + c];
    }
    {
        rm_f4 acc[4] = {};
        rm_gemm<LT_E, 4>(sX, LT_LD, a.Wr0, LT_E, wave, lane, acc);
        rm_store<4>(acc, a.br0, nullptr, 0, sB, LT_LD, 1, wave, lane);
    }
    __syncthreads();
    {
        rm_f4 acc[4] = {};
        rm_gemm<LT_E, 4>(sB, LT_LD, a.Wr2, LT_E, wave, lane, acc);
        rm_store<4>(acc, a.br2, nullptr, 0, sA, LT_LD, 1, wave, lane);
    }
    __syncthreads();
    if (wave == 0) {
        rm_f4 acc[1] = {};
        rm_gemm<LT_E, 1>(sA, LT_LD, a.Wr4, LT_E, 0, lane, acc);
        rm_store<1>(acc, a.br4, nullptr, 0, sD, 20, 0, 0, lane);
    }
    __syncthreads();
    // ---- refine + velocity + theta_d2xy: one thread per row -----------------------------------------
    if (tid < RM_ROWS && row0 + tid < n) {
        const int i = row0 + tid;
        const float *p = a.qbox + (size_t)i * 10, *d = sD + tid * 20;
        float o[10];
        o[0] = p[0] + (lt_sigmoid(d[0]) * 2.f - 1.f) / a.num_ray;
        o[1] = lt_sigmoid(d[1] + lt_inverse_sigmoid(p[1]));
        o[2] = lt_sigmoid(d[2] + lt_inverse_sigmoid(p[2]));
#pragma unroll
        for (int k = 3; k < 10; ++k)
            o[k] = d[k];
        if (a.T > 1) {
            const float td = a.td_safe[(i / a.Q) * a.T + 1];
            o[8] = o[8] / td;
            o[9] = o[9] / td;
        }
        float *pp = a.pred + (size_t)i * 10, *px = a.xy + (size_t)i * 10;
#pragma unroll
        for (int k = 0; k < 10; ++k)
            pp[k] = o[k];
        const float ang = o[0] * LT_TWO_PI, rad = o[1] * 65.0f;
        px[0] = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
        px[1] = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
#pragma unroll
        for (int k = 2; k < 10; ++k)
            px[k] = o[k];
    }
}

// The argument block is passed as an array of device pointers so the C-ABI stays plain:
//   acts[10]   : x1, bev_r, bev_l, partials, query_bbox, time_diff_safe, x3(out), cls(out), bbox_pred(out), bbox_xy(out)
//   weights[37]: in the order of TailArgs from Wor to br4
extern "C" int rac_layer_tail_fwd(const void *const *acts, const void *const *weights, int num_partials,
                                  int64_t partial_stride, int B, int Q, int T, int num_classes, int code_size,
                                  float num_ray, float eps, void *stream)
{
    RAC_CHECK_ARG(acts && weights, "rac_layer_tail_fwd: null pointer table");
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && T >= 1 && num_partials >= 1, "rac_layer_tail_fwd: bad sizes");
    RAC_CHECK_ARG(num_classes >= 1 && num_classes <= 16 && code_size == 10, "rac_layer_tail_fwd: num_classes=%d code_size=%d",
                  num_classes, code_size);
    if (B * Q == 0)
        return 0;
    for (int i = 0; i < 10; ++i)
        RAC_CHECK_ARG(acts[i] != nullptr, "rac_layer_tail_fwd: acts[%d] is null", i);
    for (int i = 0; i < LT_NUM_WEIGHTS; ++i)
        RAC_CHECK_ARG(weights[i] != nullptr, "rac_layer_tail_fwd: weights[%d] is null", i);
    TailArgs a;
    a.x1 = (const float *)acts[0]; a.bev_r = (const float *)acts[1]; a.bev_l = (const float *)acts[2];
    a.partials = (const float *)acts[3]; a.qbox = (const float *)acts[4]; a.td_safe = (const float *)acts[5];
    a.x3 = (float *)acts[6]; a.cls = (float *)acts[7]; a.pred = (float *)acts[8]; a.xy = (float *)acts[9];
    const float **w = (const float **)&a.Wor;
    static_assert(offsetof(TailArgs, br4) - offsetof(TailArgs, Wor) == (LT_NUM_WEIGHTS - 1) * sizeof(const float *),
                  "weight pointers must be contiguous in TailArgs");
    for (int i = 0; i < LT_NUM_WEIGHTS; ++i)
        w[i] = (const float *)weights[i];
    a.pstride = (long)partial_stride;
    a.n = B * Q; a.S = num_partials; a.Q = Q; a.T = T; a.num_classes = num_classes; a.code_size = code_size;
    a.eps = eps; a.num_ray = num_ray;
    const size_t lds = ((size_t)RM_ROWS * (LT_LDCAT + 4 * LT_LD) + RM_ROWS * 20) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(layer_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(layer_tail_kernel, dim3((a.n + RM_ROWS - 1) / RM_ROWS), dim3(256), lds, (hipStream_t)stream, a);
    return rac_launch_status("rac_layer_tail_fwd");
}
