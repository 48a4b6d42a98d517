// bev_pool.hip -- BEVPoolv2 (Lift-Splat-Shoot voxel pooling), forward and backward, for gfx950.
//
// Replaces the reference's other native op, models/csrc/bev_pool_v2/src/bev_pool_cuda.cu:21-136
// (bev_pool_v2_kernel / bev_pool_grad_kernel), SURVEY.md section 8 row f2:
//   forward : out[ranks_bev[s], :] = sum_{i in interval} depth[ranks_depth[i]] * feat[ranks_feat[i], :]
//   backward: intervals regrouped by ranks_feat (done by the caller, bev_pool.py:50-63):
//             depth_grad[ranks_depth[i]] = sum_c out_grad[ranks_bev[i], c] * feat[ranks_feat[i], c]
//             feat_grad[ranks_feat[s], :] = sum_{i in interval} out_grad[ranks_bev[i], :] * depth[ranks_depth[i]]
// The reference uses one thread per (interval, channel) forward and one thread per interval backward
// (a serial loop over channels).  Here a 16/32/64-lane group owns an interval with 4 channels per lane
// (16-byte loads, a whole channel row per instruction); the backward channel reduction is a DPP
// butterfly inside the group.  No atomics: every output element has exactly one writer (deterministic).
#include "rac_common.h"

// WHOLE: c is a multiple of 4 * LANES -- every lane owns a channel quad in every pass; the predicate folds away (with it in place the
// four row loads of a batch sat behind exec-mask branches of their own: 55 -> 67 us on the f8 Lift-Splat shape).
template <int LANES, bool WHOLE>
__global__ __launch_bounds__(256) void bev_pool_fwd_kernel(int c, int n_intervals, const float *__restrict__ depth,
                                                           const float *__restrict__ feat, const int *__restrict__ ranks_depth,
                                                           const int *__restrict__ ranks_feat, const int *__restrict__ ranks_bev,
                                                           const int *__restrict__ interval_starts,
                                                           const int *__restrict__ interval_lengths, float *__restrict__ out)
{
    const int grp = (blockIdx.x * 256 + threadIdx.x) / LANES;
    const int ln = threadIdx.x % LANES;
    if (grp >= n_intervals)
        return;
    const int start = interval_starts[grp], len = interval_lengths[grp];
    // Round 4: the interval is walked in chunks of LANES points whose indices and depth values the group's lanes fetch TOGETHER (one
    // coalesced round trip per chunk instead of two dependent loads per point), and the feature rows of four points are in flight
    // at once.  An interval is a serial sum (one writer per output element), and the f8 Lift-Splat frustum has intervals of up to
    // 416 points next to a mean of 26: with index -> depth -> row as a dependent chain per point the longest interval alone took
    // most of the launch (143 us for 343 MB of L2-resident rows).  Same order of additions as before.
    // Round 5: the channel pass is a loop ALL lanes of the group take (cb is group-uniform) and only the row load / store is
    // predicated on the lane's own channel quad: with `for (c0 = ln * 4; c0 < c; ...)` the lanes whose quad lies past c (any
    // c % (4 * LANES) != 0, e.g. c = 80 on 16 lanes) were masked off for the pass, never fetched their point of the chunk, and a
    // shuffle from a disabled lane reads 0 -- those points silently dropped out of the sum.
    for (int cb = 0; cb < c; cb += LANES * 4) {
        const int c0 = cb + ln * 4;
        const bool mine = WHOLE || c0 < c;
        rac_f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int base = 0; base < len; base += LANES) {
            const int n = min(LANES, len - base);
            int my_rf = 0;
            float my_d = 0.f;
            if (ln < n) {
                my_rf = ranks_feat[start + base + ln];
                my_d = depth[ranks_depth[start + base + ln]];
            }
            int i = 0;
            for (; i + 4 <= n; i += 4) {
                rac_f4 f[4];
                float d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t row = (size_t)__shfl(my_rf, i + u, LANES) * c;
                    d[u] = __shfl(my_d, i + u, LANES);
                    f[u] = mine ? rac_ld4(feat + row + c0) : rac_f4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x += f[u].x * d[u]; acc.y += f[u].y * d[u]; acc.z += f[u].z * d[u]; acc.w += f[u].w * d[u];
                }
            }
            for (; i < n; ++i) {
                const size_t row = (size_t)__shfl(my_rf, i, LANES) * c;
                const float d = __shfl(my_d, i, LANES);
                const rac_f4 f = mine ? rac_ld4(feat + row + c0) : rac_f4{0.f, 0.f, 0.f, 0.f};
                acc.x += f.x * d; acc.y += f.y * d; acc.z += f.z * d; acc.w += f.w * d;
            }
        }
        if (mine)
            *reinterpret_cast<rac_f4 *>(out + (size_t)ranks_bev[start] * c + c0) = acc;
    }
}

// scalar-channel fallback (c not a multiple of 4)
__global__ __launch_bounds__(256) void bev_pool_fwd_generic_kernel(int c, int n_intervals, const float *__restrict__ depth,
                                                                   const float *__restrict__ feat, const int *__restrict__ ranks_depth,
                                                                   const int *__restrict__ ranks_feat, const int *__restrict__ ranks_bev,
                                                                   const int *__restrict__ interval_starts,
                                                                   const int *__restrict__ interval_lengths, float *__restrict__ out)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long index = idx / c;
    const int cur_c = (int)(idx % c);
    if (index >= n_intervals)
        return;
    const int start = interval_starts[index], len = interval_lengths[index];
    float psum = 0.f;
    for (int i = 0; i < len; ++i)
        psum += feat[(size_t)ranks_feat[start + i] * c + cur_c] * depth[ranks_depth[start + i]];
    out[(size_t)ranks_bev[start] * c + cur_c] = psum;
}

template <int LANES, bool WHOLE>
__global__ __launch_bounds__(256) void bev_pool_bwd_kernel(int c, int n_intervals, const float *__restrict__ out_grad,
                                                           const float *__restrict__ depth, const float *__restrict__ feat,
                                                           const int *__restrict__ ranks_depth, const int *__restrict__ ranks_feat,
                                                           const int *__restrict__ ranks_bev, const int *__restrict__ interval_starts,
                                                           const int *__restrict__ interval_lengths, float *__restrict__ depth_grad,
                                                           float *__restrict__ feat_grad)
{
    const int grp = (blockIdx.x * 256 + threadIdx.x) / LANES;
    const int ln = threadIdx.x % LANES;
    const bool live = grp < n_intervals;
    const int start = live ? interval_starts[grp] : 0, len = live ? interval_lengths[grp] : 0;
    const int rf = live ? ranks_feat[start] : 0;
    // depth gradients: one dot product over channels per point of the interval; the points' indices are fetched by the group
    // together, a chunk of LANES at a time (as in the forward kernel), and two points' rows are in flight at once
    for (int base = 0; base < len; base += LANES) {
        const int n = min(LANES, len - base);
        int my_rb = 0, my_rd = 0;
        if (ln < n) {
            my_rb = ranks_bev[start + base + ln];
            my_rd = ranks_depth[start + base + ln];
        }
        for (int i = 0; i < n; i += 2) {
            const bool two = i + 1 < n;
            const float *og0 = out_grad + (size_t)__shfl(my_rb, i, LANES) * c;
            const float *og1 = out_grad + (size_t)__shfl(my_rb, two ? i + 1 : i, LANES) * c;
            float part0 = 0.f, part1 = 0.f;
            for (int c0 = ln * 4; c0 < c; c0 += LANES * 4) {
                const rac_f4 f = rac_ld4(feat + (size_t)rf * c + c0), g0 = rac_ld4(og0 + c0), g1 = rac_ld4(og1 + c0);
                part0 += (g0.x * f.x + g0.y * f.y) + (g0.z * f.z + g0.w * f.w);
                part1 += (g1.x * f.x + g1.y * f.y) + (g1.z * f.z + g1.w * f.w);
            }
#pragma unroll
            for (int off = LANES / 2; off >= 1; off >>= 1) {
                part0 += __shfl_xor(part0, off, LANES);
                part1 += __shfl_xor(part1, off, LANES);
            }
            const int rd0 = __shfl(my_rd, i, LANES), rd1 = __shfl(my_rd, two ? i + 1 : i, LANES);
            if (ln == 0) {
                depth_grad[rd0] = part0;
                if (two)
                    depth_grad[rd1] = part1;
            }
        }
    }
    // feature gradients: accumulated over the interval, one writer per element
    // (group-uniform channel pass, predicated row access: as in the forward kernel)
    if (live)
        for (int cb = 0; cb < c; cb += LANES * 4) {
            const int c0 = cb + ln * 4;
            const bool mine = WHOLE || c0 < c;
            rac_f4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int base = 0; base < len; base += LANES) {
                const int n = min(LANES, len - base);
                int my_rb = 0;
                float my_d = 0.f;
                if (ln < n) {
                    my_rb = ranks_bev[start + base + ln];
                    my_d = depth[ranks_depth[start + base + ln]];
                }
                int i = 0;
                for (; i + 4 <= n; i += 4) {
                    rac_f4 g[4];
                    float d[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const size_t row = (size_t)__shfl(my_rb, i + u, LANES) * c;
                        d[u] = __shfl(my_d, i + u, LANES);
                        g[u] = mine ? rac_ld4(out_grad + row + c0) : rac_f4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc.x += g[u].x * d[u]; acc.y += g[u].y * d[u]; acc.z += g[u].z * d[u]; acc.w += g[u].w * d[u];
                    }
                }
                for (; i < n; ++i) {
                    const size_t row = (size_t)__shfl(my_rb, i, LANES) * c;
                    const float d = __shfl(my_d, i, LANES);
                    const rac_f4 g = mine ? rac_ld4(out_grad + row + c0) : rac_f4{0.f, 0.f, 0.f, 0.f};
                    acc.x += g.x * d; acc.y += g.y * d; acc.z += g.z * d; acc.w += g.w * d;
                }
            }
            if (mine)
                *reinterpret_cast<rac_f4 *>(feat_grad + (size_t)rf * c + c0) = acc;
        }
}

__global__ __launch_bounds__(256) void bev_pool_bwd_generic_kernel(int c, int n_intervals, const float *__restrict__ out_grad,
                                                                   const float *__restrict__ depth, const float *__restrict__ feat,
                                                                   const int *__restrict__ ranks_depth,
                                                                   const int *__restrict__ ranks_feat, const int *__restrict__ ranks_bev,
                                                                   const int *__restrict__ interval_starts,
                                                                   const int *__restrict__ interval_lengths,
                                                                   float *__restrict__ depth_grad, float *__restrict__ feat_grad)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_intervals)
        return;
    const int start = interval_starts[idx], len = interval_lengths[idx];
    const int rf = ranks_feat[start];
    for (int i = 0; i < len; ++i) {
        const float *og = out_grad + (size_t)ranks_bev[start + i] * c;
        float s = 0.f;
        for (int cc = 0; cc < c; ++cc)
            s += og[cc] * feat[(size_t)rf * c + cc];
        depth_grad[ranks_depth[start + i]] = s;
    }
    for (int cc = 0; cc < c; ++cc) {
        float s = 0.f;
        for (int i = 0; i < len; ++i)
            s += out_grad[(size_t)ranks_bev[start + i] * c + cc] * depth[ranks_depth[start + i]];
        feat_grad[(size_t)rf * c + cc] = s;
    }
}

static int pool_lanes(int c)
{
    if (c % 4 != 0)
        return 0;
    const int q = c / 4;
    return q >= 64 ? 64 : (q >= 32 ? 32 : 16);
}

extern "C" int rac_bev_pool_v2_fwd(const float *depth, const float *feat, float *out, const int32_t *ranks_depth,
                                   const int32_t *ranks_feat, const int32_t *ranks_bev, const int32_t *interval_lengths,
                                   const int32_t *interval_starts, int c, int n_intervals, void *stream)
{
    RAC_CHECK_ARG(c >= 1 && n_intervals >= 0, "rac_bev_pool_v2_fwd: c=%d n_intervals=%d", c, n_intervals);
    if (n_intervals == 0)
        return 0;
    RAC_CHECK_ARG(depth && feat && out && ranks_depth && ranks_feat && ranks_bev && interval_lengths && interval_starts,
                  "rac_bev_pool_v2_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int lanes = pool_lanes(c);
#define POOL_FWD(L_) do { POOL_FWD2(L_, true); else POOL_FWD2(L_, false); } while (0)
#define POOL_FWD2(L_, W_) if ((c % (4 * L_) == 0) == W_) hipLaunchKernelGGL((bev_pool_fwd_kernel<L_, W_>), dim3((unsigned)(((long)n_intervals * L_ + 255) / 256)), dim3(256), 0, st, \
                                        c, n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out)
    if (lanes == 64) POOL_FWD(64);
    else if (lanes == 32) POOL_FWD(32);
    else if (lanes == 16) POOL_FWD(16);
    else
        hipLaunchKernelGGL(bev_pool_fwd_generic_kernel, dim3((unsigned)(((long)n_intervals * c + 255) / 256)), dim3(256), 0, st, c,
                           n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out);
#undef POOL_FWD
#undef POOL_FWD2
    return rac_launch_status("rac_bev_pool_v2_fwd");
}

extern "C" int rac_bev_pool_v2_bwd(const float *out_grad, float *depth_grad, float *feat_grad, const float *depth,
                                   const float *feat, const int32_t *ranks_depth, const int32_t *ranks_feat,
                                   const int32_t *ranks_bev, const int32_t *interval_lengths, const int32_t *interval_starts,
                                   int c, int n_intervals, void *stream)
{
    RAC_CHECK_ARG(c >= 1 && n_intervals >= 0, "rac_bev_pool_v2_bwd: c=%d n_intervals=%d", c, n_intervals);
    if (n_intervals == 0)
        return 0;
    RAC_CHECK_ARG(out_grad && depth_grad && feat_grad && depth && feat && ranks_depth && ranks_feat && ranks_bev &&
                      interval_lengths && interval_starts,
                  "rac_bev_pool_v2_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int lanes = pool_lanes(c);
#define POOL_BWD(L_) do { POOL_BWD2(L_, true); else POOL_BWD2(L_, false); } while (0)
#define POOL_BWD2(L_, W_) if ((c % (4 * L_) == 0) == W_) hipLaunchKernelGGL((bev_pool_bwd_kernel<L_, W_>), dim3((unsigned)(((long)n_intervals * L_ + 255) / 256)), dim3(256), 0, st, \
                                        c, n_intervals, out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,      \
                                        interval_lengths, depth_grad, feat_grad)
    if (lanes == 64) POOL_BWD(64);
    else if (lanes == 32) POOL_BWD(32);
    else if (lanes == 16) POOL_BWD(16);
    else
        hipLaunchKernelGGL(bev_pool_bwd_generic_kernel, dim3((unsigned)((n_intervals + 255) / 256)), dim3(256), 0, st, c, n_intervals,
                           out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, depth_grad,
                           feat_grad);
#undef POOL_BWD
#undef POOL_BWD2
    return rac_launch_status("rac_bev_pool_v2_bwd");
}
