// rowgemm.hip -- the small dense layers of a decoder layer (900 rows x 256..768 -> 10..2189 features) with the
// row-wise work around them folded into the GEMM launch (gfx950).
//
// A decoder layer runs ~17 such Linears, each followed or preceded by an add / split-K sum / LayerNorm / ReLU
// (models/racformer_transformer.py:170-177, 243-269).  As library GEMMs plus row-wise kernels that is ~45 launches of
// 4-14 us each -- launch- and latency-bound, a third of the layer's time.  Here one launch is
//     X   = per 256-wide segment s of the A operand:
//             [relu]( LN( a_scale * sum_p A_s[p] + bias0 + residual ) * gamma + beta ) [+ post]      (LN optional)
//     out = [relu on columns >= relu_from]( X @ W^T + b )
// i.e. the *producer's* normalisation runs as the prologue of its *consumer* GEMM.  Finished segments can be
// stored (x_out: residuals / layer outputs needed elsewhere; split_out: the f16 hi/lo image for a split-precision
// library GEMM).  Up to three independent GEMMs over the same rows share a launch (blockIdx.z).
//
// Workgroup = 16 rows x 64 output columns (grid 57 x N/64: all CUs busy at N = 256; 64 rows for N >= 512), 4 waves, one
// 16x16 tile per wave and 16 rows.  The 16 x K activation tile is built in LDS by the prologue (one wave per row, 16-byte accesses, two-pass
// statistics in registers); the weights stream from L2 straight into MFMA operands (torch's [out][in] layout, one
// 16-byte load feeds four k-steps); arithmetic is v_mfma_f32_16x16x4_f32 -- exact fp32, bit-for-bit an fmaf chain.
#include "rac_common.h"

typedef float rg_f4 __attribute__((ext_vector_type(4)));

struct RowGemmArgs {
    rac_rowgemm d[RAC_ROWGEMM_MAX_BATCH];
    int rows;
};

__device__ __forceinline__ float rg_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// NSEG: 256-wide segments of K.  MT: 16-row tiles per workgroup (1: 16 rows, all CUs busy at N = 256; 4: 64 rows, the
// weight fragments of a wave are reused by four row tiles -- a quarter of the L2 weight traffic, for the wide layers).
// EARLY: issue the first segment's weight loads above the prologue (best for the 228-workgroup launches, where one
// workgroup per CU has to hide its own latencies); without it the kernel fits 128 VGPRs = 4 workgroups per CU, which is what
// the wide layers (up to 1995 workgroups) need.
template <int NSEG, int MT, bool EARLY>
__device__ __forceinline__ void rowgemm_body(const rac_rowgemm &d, int rows, float *sX)
{
    constexpr int K = 256 * NSEG, LD = K + 4, R = 16 * MT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * R, n0 = blockIdx.y * 64;
    const bool first_slice = blockIdx.y == 0;
    // GEMM roles (wave w owns columns n0 + 16w .. +15; lane (li, lk): A row li, B column li, k = 16u + 4lk + i).  The
    // first segment's weights do not depend on the prologue: their 16 loads per lane are issued before it.  Row tiles
    // of one column slice read the same weights: each starts at a different k-block (rot) so that they do not hit the
    // same L2 lines in lock step (k is only a summation index).
    const int li = lane & 15, lk = lane >> 4;
    const int col = n0 + 16 * wave + li;
    const int wrow = col < d.N ? col : d.N - 1;
    const float *bp = d.w + (size_t)wrow * K + 4 * lk;
    const int rot = (blockIdx.x * 5) & 15;
    rac_f4 bcur[16], bnxt[16];
    if (EARLY) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            bcur[u] = rac_ld4(bp + 16 * ((u + rot) & 15));
    }

    // ---- prologue: build the R x K activation tile (wave w: rows w*R/4 .., 4 at a time; lane: 4 columns per segment) ----
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const rac_rowseg &g = d.seg[s];
        rac_f4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f}, b0 = {0.f, 0.f, 0.f, 0.f};
        if (g.gamma) {
            gm = rac_ld4(g.gamma + lane * 4);
            bt = rac_ld4(g.beta + lane * 4);
        }
        if (g.bias0)
            b0 = rac_ld4(g.bias0 + lane * 4);
#pragma unroll 1
        for (int rb = 0; rb < MT; ++rb) {
            const int rbase = (wave * MT + rb) * 4;   // first of this pass's 4 rows within the tile
            // sources first, four rows at once (and split-K partials two at a time): up to 8 independent 16-byte
            // loads in flight per lane instead of one dependent load per row and partial
            rac_f4 vv[4], rs[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = row0 + rbase + rr;
                vv[rr] = (rac_f4){0.f, 0.f, 0.f, 0.f};
                rs[rr] = (rac_f4){0.f, 0.f, 0.f, 0.f};
                if (row < rows) {
                    vv[rr] = rac_ld4(g.a + (size_t)row * g.ld_a + lane * 4);
                    if (g.residual)
                        rs[rr] = rac_ld4(g.residual + (size_t)row * g.ld_res + lane * 4);
                }
            }
            for (int p = 1; p < g.num_partials; p += 2) {
                rac_f4 w[2][4];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int row = row0 + rbase + rr;
                        w[j][rr] = (rac_f4){0.f, 0.f, 0.f, 0.f};
                        if (row < rows && p + j < g.num_partials)
                            w[j][rr] = rac_ld4(g.a + (size_t)(p + j) * g.partial_stride + (size_t)row * g.ld_a + lane * 4);
                    }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    vv[rr].x += w[0][rr].x + w[1][rr].x; vv[rr].y += w[0][rr].y + w[1][rr].y;
                    vv[rr].z += w[0][rr].z + w[1][rr].z; vv[rr].w += w[0][rr].w + w[1][rr].w;
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = rbase + rr, row = row0 + r;
                rac_f4 v = vv[rr];
                if (row < rows) {
                    v.x = v.x * g.a_scale + b0.x + rs[rr].x; v.y = v.y * g.a_scale + b0.y + rs[rr].y;
                    v.z = v.z * g.a_scale + b0.z + rs[rr].z; v.w = v.w * g.a_scale + b0.w + rs[rr].w;
                }
                if (g.gamma) {
                    const float mean = rg_wave_sum((v.x + v.y) + (v.z + v.w)) / 256.f;
                    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
                    const float rstd = 1.f / sqrtf(rg_wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 256.f + g.eps);
                    v.x = d0 * rstd * gm.x + bt.x; v.y = d1 * rstd * gm.y + bt.y;
                    v.z = d2 * rstd * gm.z + bt.z; v.w = d3 * rstd * gm.w + bt.w;
                }
                if (g.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if (g.post && row < rows) {
                    const rac_f4 w = rac_ld4(g.post + (size_t)row * g.ld_post + lane * 4);
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                *reinterpret_cast<rac_f4 *>(sX + r * LD + 256 * s + lane * 4) = v;
                if (first_slice && row < rows) {
                    if (g.x_out)
                        *reinterpret_cast<rac_f4 *>(g.x_out + (size_t)row * g.ld_xout + lane * 4) = v;
                    if (g.split_out) {   // f16 [hi | hi | lo | pad] image of v * split_scale (see rac_add_ln_fwd)
                        rac_h4 hi, lo;
                        rac_split_f16(v.x * g.split_scale, hi.x, lo.x);
                        rac_split_f16(v.y * g.split_scale, hi.y, lo.y);
                        rac_split_f16(v.z * g.split_scale, hi.z, lo.z);
                        rac_split_f16(v.w * g.split_scale, hi.w, lo.w);
                        if (g.split_layout == RAC_SPLIT_LINES) {
                            // line image [8 lines][hi 32 | lo 32] (1 KB per row): the X operand of rac_generator_fwd; a lane's 4
                            // columns sit in one line
                            _Float16 *dst = reinterpret_cast<_Float16 *>(g.split_out) + (size_t)row * 512 + (lane >> 3) * 64 + (lane & 7) * 4;
                            *reinterpret_cast<rac_h4 *>(dst) = hi;
                            *reinterpret_cast<rac_h4 *>(dst + 32) = lo;
                        } else {
                            _Float16 *dst = reinterpret_cast<_Float16 *>(g.split_out) + (size_t)row * (768 + g.split_pad);
                            *reinterpret_cast<rac_h4 *>(dst + lane * 4) = hi;
                            *reinterpret_cast<rac_h4 *>(dst + 256 + lane * 4) = hi;
                            *reinterpret_cast<rac_h4 *>(dst + 512 + lane * 4) = lo;
                            if (lane < g.split_pad)
                                dst[768 + lane] = lane < 2 ? (_Float16)g.split_scale : (_Float16)0.f;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- GEMM ----
    // weights of one 256-wide segment = 16 loads of 16 bytes per lane, all issued before the segment's MFMAs; the
    // next segment's loads are issued before the current segment's MFMAs (one memory latency per launch, not per step)
    const float *ap = sX + li * LD + 4 * lk;
    if (!EARLY) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            bcur[u] = rac_ld4(bp + 16 * ((u + rot) & 15));
    }
    rg_f4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        acc[m][0] = acc[m][1] = (rg_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sgm = 0; sgm < NSEG; ++sgm) {
        if (sgm + 1 < NSEG) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                bnxt[u] = rac_ld4(bp + 256 * (sgm + 1) + 16 * ((u + rot) & 15));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int ko = 256 * sgm + 16 * ((u + rot) & 15);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const rac_f4 a4 = *reinterpret_cast<const rac_f4 *>(ap + 16 * m * LD + ko);
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, bcur[u].x, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, bcur[u].y, acc[m][1], 0, 0, 0);
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, bcur[u].z, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, bcur[u].w, acc[m][1], 0, 0, 0);
            }
        }
        if (sgm + 1 < NSEG) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                bcur[u] = bnxt[u];
        }
    }
    const float bv = d.b ? d.b[wrow] : 0.f;
    const bool relu = col >= d.relu_from;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 16 * m + 4 * lk + r;
            float v = (acc[m][0][r] + acc[m][1][r]) + bv;
            if (relu)
                v = fmaxf(v, 0.f);
            if (row < rows && col < d.N)
                d.out[(size_t)row * d.ld_out + col] = v;
        }
}

template <int NSEG, int MT, bool EARLY>
__global__ __launch_bounds__(256, EARLY ? 2 : 4) void rowgemm_kernel(const RowGemmArgs a)
{
    extern __shared__ float smem[];
    const rac_rowgemm &d = a.d[blockIdx.z];
    if ((int)blockIdx.y * 64 >= d.N)
        return;   // (uniform per workgroup: batched GEMMs may have different widths)
    rowgemm_body<NSEG, MT, EARLY>(d, a.rows, smem);
}

extern "C" int rac_rowgemm_fwd(const rac_rowgemm *descs, int num, int rows, void *stream)
{
    RAC_CHECK_ARG(descs && num >= 1 && num <= RAC_ROWGEMM_MAX_BATCH && rows >= 0, "rac_rowgemm_fwd: num=%d rows=%d", num, rows);
    if (rows == 0)
        return 0;
    RowGemmArgs a;
    int max_n = 0, max_seg = 0;
    for (int i = 0; i < num; ++i) {
        const rac_rowgemm &d = descs[i];
        RAC_CHECK_ARG(d.num_seg >= 1 && d.num_seg <= 3 && d.N >= 1 && d.w && d.out && d.ld_out >= d.N,
                      "rac_rowgemm_fwd: GEMM %d: num_seg=%d N=%d ld_out=%d", i, d.num_seg, d.N, d.ld_out);
        for (int s = 0; s < d.num_seg; ++s) {
            const rac_rowseg &g = d.seg[s];
            RAC_CHECK_ARG(g.a && g.ld_a >= 256 && g.ld_a % 4 == 0 && g.num_partials >= 1 && g.partial_stride % 4 == 0,
                          "rac_rowgemm_fwd: GEMM %d segment %d: source rows (ld_a=%d, partials=%d)", i, s, g.ld_a, g.num_partials);
            RAC_CHECK_ARG((!g.gamma) == (!g.beta), "rac_rowgemm_fwd: GEMM %d segment %d: gamma / beta must come together", i, s);
            RAC_CHECK_ARG((!g.residual || g.ld_res % 4 == 0) && (!g.post || g.ld_post % 4 == 0) && (!g.x_out || g.ld_xout % 4 == 0),
                          "rac_rowgemm_fwd: GEMM %d segment %d: row strides must be multiples of 4", i, s);
            RAC_CHECK_ARG(!g.split_out || (g.split_pad >= 0 && g.split_pad <= 64 && g.split_pad % 4 == 0),
                          "rac_rowgemm_fwd: GEMM %d segment %d: split_pad=%d", i, s, g.split_pad);
        }
        RAC_CHECK_ARG(d.num_seg == descs[0].num_seg, "rac_rowgemm_fwd: batched GEMMs must have the same number of segments");
        a.d[i] = d;
        max_n = d.N > max_n ? d.N : max_n;
        max_seg = d.num_seg > max_seg ? d.num_seg : max_seg;
    }
    a.rows = rows;
    hipStream_t st = (hipStream_t)stream;
    // (round 5, measured and rejected: 32-row tiles (MT = 2) for the 512-wide single-segment layers -- 232 workgroups, one per CU,
    //  instead of 456 at two per CU: one plan in flight 230.8 -> 227.8 samples/s, four plans unchanged)
    const int R = 16;
    const size_t lds = (size_t)R * (256 * max_seg + 4) * sizeof(float);
    const dim3 grid((rows + R - 1) / R, (max_n + 63) / 64, num);
    if (max_seg == 1 && (long)grid.x * grid.y * grid.z > 512)
        hipLaunchKernelGGL((rowgemm_kernel<1, 1, false>), grid, dim3(256), lds, st, a);   // many workgroups: occupancy first
    else if (max_seg == 1)
        hipLaunchKernelGGL((rowgemm_kernel<1, 1, true>), grid, dim3(256), lds, st, a);
    else if (max_seg == 2)
        hipLaunchKernelGGL((rowgemm_kernel<2, 1, true>), grid, dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL((rowgemm_kernel<3, 1, true>), grid, dim3(256), lds, st, a);
    return rac_launch_status("rac_rowgemm_fwd");
}
