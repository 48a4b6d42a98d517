// add_ln.hip -- residual / split-K-sum / bias + LayerNorm (+ ReLU) as one row-wise kernel (gfx950).
//
// The decoder layer applies LayerNorm(256) ten times per layer, always right after an add
// (x + attn, x + proj, f + ffn, ...; models/racformer_transformer.py:170-177,199-205,246-258) or
// followed by a ReLU (position encoder, cls branch).  In torch each is 2-4 launches; here
//     out = [relu]( LN( sum_s a[s] + residual + bias ) * gamma + beta ) [+ post_residual]
// is one launch (input / output rows may be column slices of wider buffers: ld_a, ld_out): one wave64 per row, 16-byte loads, two-pass mean / variance in registers.
#include "rac_common.h"

#define ALN_MAX_V 4 /* float4 per lane: dim <= 1024 */

__device__ __forceinline__ float aln_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void add_ln_kernel(const float *__restrict__ a, int S, long pstride, int ld_a, float a_scale,
                                                     const float *__restrict__ residual, const float *__restrict__ bias,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     const float *__restrict__ post, float *__restrict__ out, int ld_out,
                                                     int rows, int dim, float eps, int relu,
                                                     _Float16 *__restrict__ split_out, float split_scale, int split_pad, int split_lines)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows)
        return;
    const int nv = dim >> 2;  // float4 per row
    rac_f4 x[ALN_MAX_V];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < ALN_MAX_V; ++k) {
        const int c = lane + 64 * k;
        x[k] = (rac_f4){0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            rac_f4 v = rac_ld4(a + (size_t)row * ld_a + c * 4);
            // split-K partials eight at a time: eight independent loads in flight per lane instead of one dependent load
            // per partial (32 partials of the mixing's out_proj: 4 memory latencies, not 31); the sum keeps its order
            for (int s0 = 1; s0 < S; s0 += 8) {
                rac_f4 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    w[j] = (rac_f4){0.f, 0.f, 0.f, 0.f};
                    if (s0 + j < S)
                        w[j] = rac_ld4(a + (size_t)(s0 + j) * pstride + (size_t)row * ld_a + c * 4);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v.x += w[j].x; v.y += w[j].y; v.z += w[j].z; v.w += w[j].w;
                }
            }
            v.x *= a_scale; v.y *= a_scale; v.z *= a_scale; v.w *= a_scale;
            if (bias) {
                const rac_f4 w = rac_ld4(bias + c * 4);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            if (residual) {
                const rac_f4 w = rac_ld4(residual + (size_t)row * dim + c * 4);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            x[k] = v;
            sum += (v.x + v.y) + (v.z + v.w);
        }
    }
    const float mean = aln_wave_sum(sum) / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < ALN_MAX_V; ++k)
        if (lane + 64 * k < nv) {
            const float d0 = x[k].x - mean, d1 = x[k].y - mean, d2 = x[k].z - mean, d3 = x[k].w - mean;
            sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    const float rstd = 1.f / sqrtf(aln_wave_sum(sq) / (float)dim + eps);
#pragma unroll
    for (int k = 0; k < ALN_MAX_V; ++k) {
        const int c = lane + 64 * k;
        if (c < nv) {
            const rac_f4 g = rac_ld4(gamma + c * 4), b = rac_ld4(beta + c * 4);
            rac_f4 y;
            y.x = (x[k].x - mean) * rstd * g.x + b.x;
            y.y = (x[k].y - mean) * rstd * g.y + b.y;
            y.z = (x[k].z - mean) * rstd * g.z + b.z;
            y.w = (x[k].w - mean) * rstd * g.w + b.w;
            if (relu) {
                y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
            }
            if (post) {
                const rac_f4 w = rac_ld4(post + (size_t)row * dim + c * 4);
                y.x += w.x; y.y += w.y; y.z += w.z; y.w += w.w;
            }
            *reinterpret_cast<rac_f4 *>(out + (size_t)row * ld_out + c * 4) = y;
            if (split_out) {
                // y * scale = hi + lo with hi, lo in f16 (22 significant bits together): the A operand of a
                // 3-product split GEMM on the f16 matrix cores, written K-concatenated as [hi | hi | lo]
                rac_h4 hi, lo;
                rac_split_f16(y.x * split_scale, hi.x, lo.x);
                rac_split_f16(y.y * split_scale, hi.y, lo.y);
                rac_split_f16(y.z * split_scale, hi.z, lo.z);
                rac_split_f16(y.w * split_scale, hi.w, lo.w);
                if (split_lines) {
                    // line image [dim/32 lines][hi 32 | lo 32]: the X operand of rac_generator_fwd (a lane's 4 columns sit in one line)
                    _Float16 *dl = split_out + (size_t)row * (2 * dim) + (c >> 3) * 64 + (c & 7) * 4;
                    *reinterpret_cast<rac_h4 *>(dl) = hi;
                    *reinterpret_cast<rac_h4 *>(dl + 32) = lo;
                    continue;
                }
                _Float16 *dst = split_out + (size_t)row * (3 * dim + split_pad) + c * 4;
                *reinterpret_cast<rac_h4 *>(dst) = hi;
                *reinterpret_cast<rac_h4 *>(dst + dim) = hi;
                *reinterpret_cast<rac_h4 *>(dst + 2 * dim) = lo;
                if (split_pad && c == 0) {
                    // bias columns of the K-concatenated GEMM: a constant activation 1.0 (scaled like the others)
                    // against [bias_hi | bias_lo] rows of the weight image; the rest of the pad is zero
                    _Float16 *pad = split_out + (size_t)row * (3 * dim + split_pad) + 3 * dim;
                    for (int i = 0; i < split_pad; ++i)
                        pad[i] = i < 2 ? (_Float16)split_scale : (_Float16)0.f;
                }
            }
        }
    }
}

// Many split-K partials (the 32 slices of the mixing's out_proj): one row per workgroup, its four waves sum a quarter of the
// partials each (eight independent 16-byte loads in flight per lane), wave 0 adds the quarters in a fixed order and normalises.
// 900 workgroups instead of 225: four times the loads in flight for the 29.5 MB of partials.  dim == 256 only.
__global__ __launch_bounds__(256) void add_ln_many_kernel(const float *__restrict__ a, int S, long pstride, int ld_a, float a_scale,
                                                          const float *__restrict__ residual, const float *__restrict__ bias,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ post, float *__restrict__ out, int ld_out,
                                                          int rows, float eps, int relu,
                                                          _Float16 *__restrict__ split_out, float split_scale, int split_pad, int split_lines)
{
    __shared__ rac_f4 part[3][64];
    const int row = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (S + 3) / 4, s_lo = wave * per, s_hi = min(S, s_lo + per);
    rac_f4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
        rac_f4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w[j] = (rac_f4){0.f, 0.f, 0.f, 0.f};
            if (s0 + j < s_hi)
                w[j] = rac_ld4(a + (size_t)(s0 + j) * pstride + (size_t)row * ld_a + lane * 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v.x += w[j].x; v.y += w[j].y; v.z += w[j].z; v.w += w[j].w;
        }
    }
    if (wave > 0)
        part[wave - 1][lane] = v;
    __syncthreads();
    if (wave > 0)
        return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const rac_f4 w = part[k][lane];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    v.x *= a_scale; v.y *= a_scale; v.z *= a_scale; v.w *= a_scale;
    if (bias) {
        const rac_f4 w = rac_ld4(bias + lane * 4);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (residual) {
        const rac_f4 w = rac_ld4(residual + (size_t)row * 256 + lane * 4);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    const float mean = aln_wave_sum((v.x + v.y) + (v.z + v.w)) / 256.f;
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float rstd = 1.f / sqrtf(aln_wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 256.f + eps);
    const rac_f4 g = rac_ld4(gamma + lane * 4), b = rac_ld4(beta + lane * 4);
    rac_f4 y = {d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w};
    if (relu) {
        y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
    }
    if (post) {
        const rac_f4 w = rac_ld4(post + (size_t)row * 256 + lane * 4);
        y.x += w.x; y.y += w.y; y.z += w.z; y.w += w.w;
    }
    *reinterpret_cast<rac_f4 *>(out + (size_t)row * ld_out + lane * 4) = y;
    if (split_out) {
        rac_h4 hi, lo;
        rac_split_f16(y.x * split_scale, hi.x, lo.x);
        rac_split_f16(y.y * split_scale, hi.y, lo.y);
        rac_split_f16(y.z * split_scale, hi.z, lo.z);
        rac_split_f16(y.w * split_scale, hi.w, lo.w);
        if (split_lines) {
            _Float16 *dl = split_out + (size_t)row * 512 + (lane >> 3) * 64 + (lane & 7) * 4;
            *reinterpret_cast<rac_h4 *>(dl) = hi;
            *reinterpret_cast<rac_h4 *>(dl + 32) = lo;
        } else {
            _Float16 *dst = split_out + (size_t)row * (768 + split_pad) + lane * 4;
            *reinterpret_cast<rac_h4 *>(dst) = hi;
            *reinterpret_cast<rac_h4 *>(dst + 256) = hi;
            *reinterpret_cast<rac_h4 *>(dst + 512) = lo;
            if (split_pad && lane == 0) {
                _Float16 *pad = split_out + (size_t)row * (768 + split_pad) + 768;
                for (int i = 0; i < split_pad; ++i)
                    pad[i] = i < 2 ? (_Float16)split_scale : (_Float16)0.f;
            }
        }
    }
}

// position-encoder head: out = relu(LN(W x + b)) for a 3-wide input (models/racformer_transformer.py:170-173);
// a GEMM with K=3 is pure launch overhead, so the three FMAs per output are done here.
__global__ __launch_bounds__(256) void pe_head_kernel(const float *__restrict__ x, int ld_x, const float *__restrict__ W,
                                                      const float *__restrict__ bias, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float *__restrict__ out, int rows, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows)
        return;
    const float x0 = x[(size_t)row * ld_x], x1 = x[(size_t)row * ld_x + 1], x2 = x[(size_t)row * ld_x + 2];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + j;
        v[j] = W[c * 3] * x0 + W[c * 3 + 1] * x1 + W[c * 3 + 2] * x2 + bias[c];
    }
    const float mean = aln_wave_sum((v[0] + v[1]) + (v[2] + v[3])) / 256.f;
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    const float rstd = 1.f / sqrtf(aln_wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 256.f + eps);
    const rac_f4 g = rac_ld4(gamma + lane * 4), b = rac_ld4(beta + lane * 4);
    rac_f4 y = {fmaxf(d0 * rstd * g.x + b.x, 0.f), fmaxf(d1 * rstd * g.y + b.y, 0.f), fmaxf(d2 * rstd * g.z + b.z, 0.f),
                fmaxf(d3 * rstd * g.w + b.w, 0.f)};
    *reinterpret_cast<rac_f4 *>(out + (size_t)row * 256 + lane * 4) = y;
}

extern "C" int rac_add_ln_fwd(const float *a, int num_partials, int64_t partial_stride, int ld_a, float a_scale, const float *residual,
                              const float *bias, const float *gamma, const float *beta, const float *post_residual,
                              float *out, int ld_out, int rows, int dim, float eps, int relu, void *split_out,
                              float split_scale, int split_pad, int split_layout, void *stream)
{
    RAC_CHECK_ARG(split_layout == RAC_SPLIT_KCAT || (split_layout == RAC_SPLIT_LINES && dim % 32 == 0 && split_pad == 0),
                  "rac_add_ln_fwd: split_layout=%d (RAC_SPLIT_LINES needs dim %% 32 == 0 and split_pad == 0)", split_layout);
    RAC_CHECK_ARG(rows >= 0 && dim >= 4 && dim % 4 == 0 && dim <= 256 * ALN_MAX_V, "rac_add_ln_fwd: dim=%d (multiple of 4, <= %d)", dim, 256 * ALN_MAX_V);
    RAC_CHECK_ARG(num_partials >= 1, "rac_add_ln_fwd: num_partials=%d", num_partials);
    RAC_CHECK_ARG(!split_out || split_pad == 0 || (split_pad >= 2 && split_pad % 4 == 0), "rac_add_ln_fwd: split_pad=%d (0, or a multiple of 4)", split_pad);
    RAC_CHECK_ARG(ld_a >= dim && ld_out >= dim && ld_a % 4 == 0 && ld_out % 4 == 0, "rac_add_ln_fwd: row strides ld_a=%d ld_out=%d", ld_a, ld_out);
    if (rows == 0)
        return 0;
    RAC_CHECK_ARG(a && gamma && beta && out, "rac_add_ln_fwd: null pointer");
    if (dim == 256 && num_partials >= 8) {
        hipLaunchKernelGGL(add_ln_many_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, num_partials, (long)partial_stride, ld_a,
                           a_scale, residual, bias, gamma, beta, post_residual, out, ld_out, rows, eps, relu,
                           reinterpret_cast<_Float16 *>(split_out), split_scale, split_pad, split_layout == RAC_SPLIT_LINES ? 1 : 0);
        return rac_launch_status("rac_add_ln_fwd");
    }
    hipLaunchKernelGGL(add_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, num_partials,
                       (long)partial_stride, ld_a, a_scale, residual, bias, gamma, beta, post_residual, out, ld_out, rows, dim, eps, relu,
                       reinterpret_cast<_Float16 *>(split_out), split_scale, split_pad, split_layout == RAC_SPLIT_LINES ? 1 : 0);
    return rac_launch_status("rac_add_ln_fwd");
}

extern "C" int rac_pe_head_fwd(const float *x, int ld_x, const float *weight, const float *bias, const float *gamma,
                               const float *beta, float *out, int rows, int dim, float eps, void *stream)
{
    RAC_CHECK_ARG(dim == 256 && rows >= 0 && ld_x >= 3, "rac_pe_head_fwd: dim=%d (built for 256), ld_x=%d", dim, ld_x);
    if (rows == 0)
        return 0;
    RAC_CHECK_ARG(x && weight && bias && gamma && beta && out, "rac_pe_head_fwd: null pointer");
    hipLaunchKernelGGL(pe_head_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld_x, weight, bias, gamma,
                       beta, out, rows, eps);
    return rac_launch_status("rac_pe_head_fwd");
}
