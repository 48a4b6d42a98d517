// layer_tail.hip -- everything of a decoder layer behind AdaptiveMixing's out_proj as ONE launch (gfx950):
//   BEV output projections (x2) -> norm_radar_bev / norm_lss_bev (+ identity) -> fusion Linear(768 -> 256) -> norm_fusion -> FFN
//   (256 -> 512 -> 256, + identity) -> norm3 -> cls_branch (Linear, LN, ReLU, Linear, LN, ReLU, Linear) || reg_branch (Linear, ReLU,
//   Linear, ReLU, Linear)                                  (models/racformer_transformer.py:248-262, bev_self_attention.py:215-225)
//
// Rounds 2-4 ran this as seven rac_rowgemm_fwd launches (producer's LayerNorm in the consumer GEMM's prologue): 8-14 us each for
// about 1 us of arithmetic -- every launch pays its own ramp, one or two dependent memory round trips and a drain, 87 us per layer
// in all (profiles/r04_timeline_one_replay.txt).  All of these stages are ROW-wise, so here one workgroup carries a 16-row tile
// through all of them: activations never leave LDS between stages, the only global traffic is the weights (3.3 MB per layer,
// L2-resident, streamed once per row tile straight into MFMA operands) and the layer's outputs.
// What bounds it: a row tile needs EVERY weight of the chain, i.e. 3.3 MB through one CU's L2 port (~66 GB/s: 50 us), and 13
// 16 x 256 x 256 products on one CU's fp32 matrix pipes (v_mfma_f32_16x16x4_f32, exact fp32 as in rowgemm.hip: 4 us each) -- the
// two overlap.  57 workgroups: the other 199 CUs stay free for the samples in flight beside this one.
//
// Workgroup = 512 threads = 8 waves, 16 rows.  Phases alternate between ROW OPS (wave w: rows 2w, 2w+1; lane: 4 columns of a
// 256-wide segment; LayerNorm statistics by wave reduction, two-pass as rowgemm.hip) and GEMMs (the 16-column tiles of the phase's
// GEMMs dealt round-robin to the waves; per tile and 256-deep K segment 16 weight loads of 16 bytes per lane, the next item's
// loads in flight under the current item's 64 MFMAs), one workgroup barrier between phases.
#include "rac_common.h"

typedef float lt_f4 __attribute__((ext_vector_type(4)));

#define LT_ROWS 16
#define LT_LDA (768 + 4)      /* A operand buffer: up to K = 768 */
#define LT_LDT (512 + 4)      /* two result buffers of up to 512 columns */
#define LT_LDF (256 + 4)      /* the fusion output kept for the FFN's identity */
#define LT_LDS_FLOATS (LT_ROWS * (LT_LDA + 2 * LT_LDT + LT_LDF))

__device__ __forceinline__ float lt_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// LayerNorm over the 256 columns of a row held as 4 values per lane (two-pass statistics, as rowgemm.hip / add_ln.hip)
__device__ __forceinline__ rac_f4 lt_ln(rac_f4 v, const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int lane)
{
    const float mean = lt_wave_sum((v.x + v.y) + (v.z + v.w)) / 256.f;
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float rstd = 1.f / sqrtf(lt_wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 256.f + eps);
    const rac_f4 g = rac_ld4(gamma + lane * 4), b = rac_ld4(beta + lane * 4);
    return rac_f4{d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w};
}
__device__ __forceinline__ rac_f4 lt_relu(rac_f4 v)
{
    return rac_f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}
__device__ __forceinline__ rac_f4 lt_add(rac_f4 a, rac_f4 b)
{
    return rac_f4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
}

// One GEMM of a phase: out[r][c] = [relu from column relu_from](sum_k A[r][k] W[c][k] + b[c]), A in LDS, W [N][ldw] in global memory.
struct LtGemm {
    const float *a;       // LDS, row stride lda, K columns from here
    int lda;
    const float *w;       // [N][ldw]
    const float *b;       // [N] or null
    int ldw, N, relu_from;
    float *dst;           // LDS (ldd = row stride) or, with dst_global, global [rows][ldd]
    int ldd;
    bool dst_global;
};

// NSEG 256-deep K segments; up to two GEMMs side by side (same K): their 16-column tiles are numbered through and dealt to the waves.
// A wave's work is a sequence of ITEMS (tile, K segment): 16 weight loads of 16 bytes per lane, 64 MFMAs.
// Loads are issued in LINE order -- lane = 4 * (weight row of the tile) + (16-byte piece): a quad of lanes reads 64 contiguous bytes --
// and brought into fragment order (lane = 16 * piece + row) by ds_bpermute.  In fragment order four consecutive lanes read four
// different weight rows, the texture addresser sees 64 separate requests per instruction, and this kernel -- whose 57 workgroups
// each pull the whole 3.3 MB of weights through one CU -- ran at 136 us per layer instead of 87 for the seven launches it replaces.
// Pipeline per item: the NEXT item's 16 raw loads are issued first; after the four MFMAs that consume bcur[u], bcur[u] is overwritten
// by the permuted raw[u] of the next item (an MFMA reads its operands at issue; the permute lands 60 MFMAs before it is needed).
template <int NSEG>
__device__ __forceinline__ void lt_gemm_phase(const LtGemm &g0, const LtGemm &g1, int ngemm, int row0, int rows, int wave, int lane)
{
    const int li = lane & 15, lk = lane >> 4;
    const int lrow = lane >> 2, lq = lane & 3;                  // loader role: weight row of the tile, 16-byte piece of a 64-byte group
    const int perm = (4 * li + lk) * 4;                         // fragment lane (li, lk) <- loader lane 4 li + lk
    const int t0 = (g0.N + 15) >> 4, t1 = ngemm > 1 ? (g1.N + 15) >> 4 : 0;
    const int ntiles = t0 + t1;
    int tile = wave;
    if (tile >= ntiles)
        return;
    auto wptr = [&](int t) {
        const LtGemm &g = t < t0 ? g0 : g1;
        const int col = (t < t0 ? t : t - t0) * 16 + lrow;
        return g.w + (size_t)(col < g.N ? col : g.N - 1) * g.ldw + 4 * lq;
    };
    rac_f4 bcur[16], raw[16];
    auto permute = [&](const rac_f4 &r) {
        rac_f4 o;
        o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(perm, __float_as_int(r.x)));
        o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(perm, __float_as_int(r.y)));
        o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(perm, __float_as_int(r.z)));
        o.w = __int_as_float(__builtin_amdgcn_ds_bpermute(perm, __float_as_int(r.w)));
        return o;
    };
    const float *bp = wptr(tile);
#pragma unroll
    for (int u = 0; u < 16; ++u)
        raw[u] = rac_ld4(bp + 16 * u);
#pragma unroll
    for (int u = 0; u < 16; ++u)
        bcur[u] = permute(raw[u]);
    for (; tile < ntiles; tile += 8) {
        const LtGemm &g = tile < t0 ? g0 : g1;
        const int col = (tile < t0 ? tile : tile - t0) * 16 + li;
        const float *ap = g.a + li * g.lda + 4 * lk;
        const float *bpn = tile + 8 < ntiles ? wptr(tile + 8) : bp;      // (past the end: harmless re-read of this tile's weights)
        lt_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const float *nx = s + 1 < NSEG ? bp + 256 * (s + 1) : bpn;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                raw[u] = rac_ld4(nx + 16 * u);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const rac_f4 a4 = *reinterpret_cast<const rac_f4 *>(ap + 256 * s + 16 * u);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, bcur[u].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, bcur[u].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, bcur[u].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, bcur[u].w, acc1, 0, 0, 0);
                bcur[u] = permute(raw[u]);
            }
        }
        bp = bpn;
        const bool live = col < g.N;
        const float bv = (g.b && live) ? g.b[col] : 0.f;
        const bool relu = col >= g.relu_from;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * lk + r;
            float v = (acc0[r] + acc1[r]) + bv;
            if (relu)
                v = fmaxf(v, 0.f);
            if (g.dst_global) {
                if (live && row0 + row < rows)
                    g.dst[(size_t)(row0 + row) * g.ldd + col] = v;
            } else if (live) {
                g.dst[row * g.ldd + col] = v;
            }
        }
    }
}

struct LayerTailArgs {
    rac_layer_tail p;
    int rows;
};

// Diagnostic builds only (tools/build_variant.sh ltstamps "-DLT_STAMPS" layer_tail.hip; tools/layer_tail_phases.py): wave 0 of every
// workgroup stamps s_memtime at each phase boundary.  Not part of the product library or its ABI.
#if defined(LT_STAMPS) && defined(RAC_DIAGNOSTIC_BUILD)
__device__ unsigned long long lt_stamp_buf[64 * 16];
#define LT_STAMP(i_)                                                                                   \
    do {                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 64)                                                       \
            lt_stamp_buf[blockIdx.x * 16 + (i_)] = __builtin_amdgcn_s_memtime();                       \
    } while (0)
extern "C" int rac_dbg_layer_tail_stamps(unsigned long long *host_out)
{
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lt_stamp_buf), sizeof(unsigned long long) * 64 * 16, 0, hipMemcpyDeviceToHost);
}
#else
#define LT_STAMP(i_)
#endif

__global__ __launch_bounds__(512, 1) void layer_tail_kernel(const LayerTailArgs a)
{
    extern __shared__ float lt_lds[];
    const rac_layer_tail &p = a.p;
    float *sA = lt_lds, *sT0 = sA + LT_ROWS * LT_LDA, *sT1 = sT0 + LT_ROWS * LT_LDT, *sF = sT1 + LT_ROWS * LT_LDT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * LT_ROWS, rows = a.rows;
    const int c4 = lane * 4;
    const rac_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](const float *base, int ld, int row) { return row0 + row < rows ? rac_ld4(base + (size_t)(row0 + row) * ld + c4) : zero4; };
    auto gstore = [&](float *base, int ld, int row, rac_f4 v) {
        if (base && row0 + row < rows)
            *reinterpret_cast<rac_f4 *>(base + (size_t)(row0 + row) * ld + c4) = v;
    };
    auto lds4 = [&](float *buf, int ld, int row, int coff) { return reinterpret_cast<rac_f4 *>(buf + row * ld + coff + c4); };
    const LtGemm none = {};

    LT_STAMP(0);
    // ---- phase 0: the two BEV streams' rows -> A[:, 0:512]
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        *lds4(sA, LT_LDA, r, 0) = gload(p.bev, 256, r);
        *lds4(sA, LT_LDA, r, 256) = gload(p.bev + (size_t)p.bev_stream_stride, 256, r);
    }
    __syncthreads();
    LT_STAMP(1);
    // ---- phase 1: output_proj of both streams (bev_self_attention.py:215) -> T0[:, 0:256 | 256:512]
    {
        const LtGemm gr = {sA, LT_LDA, p.bev_w[0], p.bev_b[0], 256, 256, 1 << 30, sT0, LT_LDT, false};
        const LtGemm gl = {sA + 256, LT_LDA, p.bev_w[1], p.bev_b[1], 256, 256, 1 << 30, sT0 + 256, LT_LDT, false};
        lt_gemm_phase<1>(gr, gl, 2, row0, rows, wave, lane);
    }
    __syncthreads();
    LT_STAMP(2);
    // ---- phase 2: A = [norm2 output | norm_radar_bev(proj_r + x1) | norm_lss_bev(proj_l + x1)]  (racformer_transformer.py:248-256)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        const rac_f4 x1 = gload(p.x1, 256, r);
        *lds4(sA, LT_LDA, r, 0) = gload(p.x2, 256, r);
        const rac_f4 pr = lt_add(*lds4(sT0, LT_LDT, r, 0), x1), pl = lt_add(*lds4(sT0, LT_LDT, r, 256), x1);
        gstore(p.probe_radar, 256, r, pr);
        gstore(p.probe_lss, 256, r, pl);
        *lds4(sA, LT_LDA, r, 256) = lt_ln(pr, p.nr_g, p.nr_b, p.eps, lane);
        *lds4(sA, LT_LDA, r, 512) = lt_ln(pl, p.nl_g, p.nl_b, p.eps, lane);
    }
    __syncthreads();
    LT_STAMP(3);
    // ---- phase 3: fusion Linear (K = 768) -> T1[:, 0:256]
    {
        const LtGemm g = {sA, LT_LDA, p.fus_w, p.fus_b, 768, 256, 1 << 30, sT1, LT_LDT, false};
        lt_gemm_phase<3>(g, none, 1, row0, rows, wave, lane);
    }
    __syncthreads();
    LT_STAMP(4);
    // ---- phase 4: f = norm_fusion(.) -> F (kept for the FFN's identity) and A[:, 0:256]
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        const rac_f4 f = lt_ln(*lds4(sT1, LT_LDT, r, 0), p.nf_g, p.nf_b, p.eps, lane);
        *lds4(sF, LT_LDF, r, 0) = f;
        *lds4(sA, LT_LDA, r, 0) = f;
    }
    __syncthreads();
    LT_STAMP(5);
    // ---- phase 5: FFN layer 1 + ReLU -> T0[:, 0:512]
    {
        const LtGemm g = {sA, LT_LDA, p.ffn1_w, p.ffn1_b, 256, 512, 0, sT0, LT_LDT, false};
        lt_gemm_phase<1>(g, none, 1, row0, rows, wave, lane);
    }
    __syncthreads();
    LT_STAMP(6);
    // ---- phase 6: FFN layer 2 (K = 512, straight from T0) -> T1[:, 0:256]
    {
        const LtGemm g = {sT0, LT_LDT, p.ffn2_w, p.ffn2_b, 512, 256, 1 << 30, sT1, LT_LDT, false};
        lt_gemm_phase<2>(g, none, 1, row0, rows, wave, lane);
    }
    __syncthreads();
    LT_STAMP(7);
    // ---- phase 7: x3 = norm3(f + ffn) -> the layer's output features and A[:, 0:256]
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        const rac_f4 s = lt_add(*lds4(sF, LT_LDF, r, 0), *lds4(sT1, LT_LDT, r, 0));
        gstore(p.probe_ffn, 256, r, s);
        const rac_f4 x3 = lt_ln(s, p.n3_g, p.n3_b, p.eps, lane);
        gstore(p.x3_out, 256, r, x3);
        *lds4(sA, LT_LDA, r, 0) = x3;
    }
    __syncthreads();
    LT_STAMP(8);
    // ---- phase 8: first Linear of both branches as one 256 -> 512 GEMM (ReLU on the reg half) -> T0[:, 0:512]
    {
        const LtGemm g = {sA, LT_LDA, p.c0r0_w, p.c0r0_b, 256, 512, 256, sT0, LT_LDT, false};
        lt_gemm_phase<1>(g, none, 1, row0, rows, wave, lane);
    }
    __syncthreads();
    LT_STAMP(9);
    // ---- phase 9: cls: LN + ReLU -> A[:, 0:256]  (the reg half is used as it lies in T0)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        *lds4(sA, LT_LDA, r, 0) = lt_relu(lt_ln(*lds4(sT0, LT_LDT, r, 0), p.c1_g, p.c1_b, p.eps, lane));
    }
    __syncthreads();
    LT_STAMP(10);
    // ---- phase 10: cls_branch[3] || reg_branch[2] (+ ReLU) -> T1[:, 0:256 | 256:512]
    {
        const LtGemm gc = {sA, LT_LDA, p.c3_w, p.c3_b, 256, 256, 1 << 30, sT1, LT_LDT, false};
        const LtGemm gr = {sT0 + 256, LT_LDT, p.r2_w, p.r2_b, 256, 256, 0, sT1 + 256, LT_LDT, false};
        lt_gemm_phase<1>(gc, gr, 2, row0, rows, wave, lane);
    }
    __syncthreads();
    LT_STAMP(11);
    // ---- phase 11: cls: LN + ReLU -> A[:, 0:256]
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        *lds4(sA, LT_LDA, r, 0) = lt_relu(lt_ln(*lds4(sT1, LT_LDT, r, 0), p.c4_g, p.c4_b, p.eps, lane));
    }
    __syncthreads();
    LT_STAMP(12);
    // ---- phase 12: the two narrow output Linears -> class logits and box deltas (global)
    {
        const LtGemm gc = {sA, LT_LDA, p.c6_w, p.c6_b, 256, p.num_classes, 1 << 30, p.cls_out, p.num_classes, true};
        const LtGemm gr = {sT1 + 256, LT_LDT, p.r4_w, p.r4_b, 256, p.code_size, 1 << 30, p.delta_out, p.code_size, true};
        lt_gemm_phase<1>(gc, gr, 2, row0, rows, wave, lane);
    }
    LT_STAMP(13);
}

extern "C" int rac_layer_tail_fwd(const rac_layer_tail *p, int rows, void *stream)
{
    RAC_CHECK_ARG(p && rows >= 0, "rac_layer_tail_fwd: null descriptor / rows=%d", rows);
    if (rows == 0)
        return 0;
    RAC_CHECK_ARG(p->num_classes >= 1 && p->num_classes <= 256 && p->code_size >= 1 && p->code_size <= 256 && p->eps > 0.f,
                  "rac_layer_tail_fwd: num_classes=%d code_size=%d", p->num_classes, p->code_size);
    const void *need[] = {p->bev, p->x1, p->x2, p->bev_w[0], p->bev_w[1], p->nr_g, p->nr_b, p->nl_g, p->nl_b, p->fus_w, p->nf_g, p->nf_b, p->ffn1_w,
                          p->ffn2_w, p->n3_g, p->n3_b, p->c0r0_w, p->c1_g, p->c1_b, p->c3_w, p->c4_g, p->c4_b, p->c6_w, p->r2_w, p->r4_w,
                          p->x3_out, p->cls_out, p->delta_out};
    uintptr_t bits = 0;
    for (const void *q : need) {
        RAC_CHECK_ARG(q, "rac_layer_tail_fwd: null pointer");
        bits |= reinterpret_cast<uintptr_t>(q);
    }
    for (const void *q : {(const void *)p->bev_b[0], (const void *)p->bev_b[1], (const void *)p->fus_b, (const void *)p->ffn1_b, (const void *)p->ffn2_b,
                          (const void *)p->c0r0_b, (const void *)p->c3_b, (const void *)p->r2_b, (const void *)p->probe_radar,
                          (const void *)p->probe_lss, (const void *)p->probe_ffn})
        bits |= reinterpret_cast<uintptr_t>(q);
    RAC_CHECK_ARG((bits & 15) == 0 && p->bev_stream_stride % 4 == 0, "rac_layer_tail_fwd: pointers must be 16-byte aligned");
    LayerTailArgs a;
    a.p = *p;
    a.rows = rows;
    const int lds = LT_LDS_FLOATS * (int)sizeof(float);
    if (const int rc = rac_set_dynamic_lds_once(RAC_ATTR_LAYER_TAIL, reinterpret_cast<const void *>(layer_tail_kernel), lds))
        return rc;
    hipLaunchKernelGGL(layer_tail_kernel, dim3((unsigned)((rows + LT_ROWS - 1) / LT_ROWS)), dim3(512), lds, (hipStream_t)stream, a);
    return rac_launch_status("rac_layer_tail_fwd");
}
