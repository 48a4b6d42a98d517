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
// Workgroup = 16 rows x 64 output columns (grid 57 x N/64: all CUs busy at N = 256), 4 waves, one 16x16 tile per
// wave.  The 16 x K activation tile is built in LDS by the prologue (one wave per row, 16-byte accesses, two-pass
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

template <int NSEG>
__device__ __forceinline__ void rowgemm_body(const rac_rowgemm &d, int rows, float *sX)
{
    constexpr int K = 256 * NSEG, LD = K + 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * 16, n0 = blockIdx.y * 64;
    const bool first_slice = blockIdx.y == 0;

    // ---- prologue: build the 16 x K activation tile (wave w: rows 4w..4w+3, lane: 4 columns of each segment) ----
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
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr, row = row0 + r;
            rac_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < rows) {
                const float *ap = g.a + (size_t)row * g.ld_a + lane * 4;
                v = rac_ld4(ap);
                for (int p = 1; p < g.num_partials; ++p) {
                    const rac_f4 w = rac_ld4(ap + (size_t)p * g.partial_stride);
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                v.x = v.x * g.a_scale + b0.x; v.y = v.y * g.a_scale + b0.y;
                v.z = v.z * g.a_scale + b0.z; v.w = v.w * g.a_scale + b0.w;
                if (g.residual) {
                    const rac_f4 w = rac_ld4(g.residual + (size_t)row * g.ld_res + lane * 4);
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
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
    __syncthreads();

    // ---- GEMM: wave w owns columns n0 + 16w .. +15;  lane (li, lk): A row li, B column li, k = 16u + 4lk + i ----
    const int li = lane & 15, lk = lane >> 4;
    const int col = n0 + 16 * wave + li;
    const int wrow = col < d.N ? col : d.N - 1;
    const float *ap = sX + li * LD + 4 * lk;
    const float *bp = d.w + (size_t)wrow * K + 4 * lk;
    rg_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = K / 16;
#pragma unroll 8
    for (int u = 0; u < U; ++u) {
        const rac_f4 a4 = *reinterpret_cast<const rac_f4 *>(ap + 16 * u);
        const rac_f4 b4 = rac_ld4(bp + 16 * u);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc1, 0, 0, 0);
    }
    const float bv = d.b ? d.b[wrow] : 0.f;
    const bool relu = col >= d.relu_from;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * lk + r;
        float v = (acc0[r] + acc1[r]) + bv;
        if (relu)
            v = fmaxf(v, 0.f);
        if (row < rows && col < d.N)
            d.out[(size_t)row * d.ld_out + col] = v;
    }
}

__global__ __launch_bounds__(256) void rowgemm_kernel(const RowGemmArgs a)
{
    extern __shared__ float smem[];
    const rac_rowgemm &d = a.d[blockIdx.z];
    if ((int)blockIdx.y * 64 >= d.N)
        return;   // (uniform per workgroup: batched GEMMs may have different widths)
    if (d.num_seg == 1)
        rowgemm_body<1>(d, a.rows, smem);
    else if (d.num_seg == 2)
        rowgemm_body<2>(d, a.rows, smem);
    else
        rowgemm_body<3>(d, a.rows, smem);
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
        a.d[i] = d;
        max_n = d.N > max_n ? d.N : max_n;
        max_seg = d.num_seg > max_seg ? d.num_seg : max_seg;
    }
    a.rows = rows;
    const size_t lds = (size_t)16 * (256 * max_seg + 4) * sizeof(float);
    hipLaunchKernelGGL(rowgemm_kernel, dim3((rows + 15) / 16, (max_n + 63) / 64, num), dim3(256), lds, (hipStream_t)stream, a);
    return rac_launch_status("rac_rowgemm_fwd");
}
