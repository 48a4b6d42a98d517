// value_proj.hip -- the value projection of a BEV stream (nn.Linear(256 -> 256) over every pixel of every frame,
// models/bev_self_attention.py:162-174) on the f16 matrix cores, straight from the channel-first maps (gfx950).
//
//   out[f*HW + p][n] = sum_c x[f][c][p] * W[n][c] + add[p][n]            (add: value_proj(pos) + bias, frame-independent)
//
// 131072 pixels x 256 x 256 per stream: as a library call this was an fp32 GEMM with a transposed operand plus a
// broadcast copy of the additive term (170 + 29 us).  Here the [C][H*W] -> [H*W][C] transpose, the hi / lo split of the
// activations and the additive term are part of the GEMM: the maps are read once (fp32, 128-byte segments), the values
// written once.  HBM-bound: 2 x 134 MB.
//
// Arithmetic as in gemm_split.hip: x * 2^e = hi + lo (two f16), W image [256][8 lines][hi 32 | lo 32] from
// rac_gemm_split_pack_fwd, three MFMA products per K step, fp32 accumulate.  The activation scale 2^e is chosen PER PIXEL
// (max over the pixel's 256 channels -> largest value in [2^13, 2^14)), so no pass over the maps for a global maximum is
// needed and small-magnitude pixels keep their full 22 bits; the epilogue multiplies each row by its own 2^-e.
//
// Workgroup = 512 threads, 8 waves, one per CU: wave w keeps the weights of features 32w .. 32w+31 in registers (as
// generator_ws_kernel) and walks `pixels_per_wg` pixels in stages of 32.  Per stage, wave w also loads channels
// 32w .. 32w+31 of the 32 pixels (lane = pixel quad x channel quad: 4 x float4), the per-pixel maxima are combined across
// the waves through LDS, and the scaled hi / lo values go to LDS in the fragment layout of gemm_split.hip (1 KB per pixel,
// 16-byte chunk c at slot c ^ (pixel & 15)).  The next stage's loads are issued before the stage's MFMAs.
#include "rac_common.h"

typedef _Float16 vp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 vp_h4 __attribute__((ext_vector_type(4)));
typedef float vp_f4 __attribute__((ext_vector_type(4)));

#define VP_ROWS 32
#define VP_BUF (VP_ROWS * 1024 + VP_ROWS * 4)   /* X image of a stage + its per-row output scales */
#define VP_LDS (2 * VP_BUF + 8 * VP_ROWS * 4)   /* two buffers + the waves' partial maxima */
#define VP_LDS_Q16 (VP_LDS + 8 * VP_ROWS * 4)   /* + the waves' partial block maxima of the q16 epilogue */

struct ValueProjArgs {
    const float *x;      // [F][256][HW]
    const char *w;       // W image [256][8][hi 32 | lo 32] f16
    const float *add;    // [HW][256] or null
    const float *bias;   // [256] or null (used when add is null)
    float *out;          // [F*HW][256]
    short *q;            // QOUT: [F*HW][256] int16 mantissas and
    float *qscale;       //   [F*HW][4] one power-of-two scale per (pixel, 64-feature block) -- quant.hip's storage; out unused
    float w_alpha;       // 2^-s of the weight image
    int HW;
    long M;              // F * HW
    int pixels_per_wg;   // multiple of VP_ROWS
};

// QOUT: the result leaves in the int16 block storage of quant.hip (what rac_bev_sampling_multi_q16_fwd reads), bit for bit what
// rac_quant_i16_fwd makes of the fp32 output: a (pixel, head) block of 64 features is held by two waves (32 features each), so
// the block maximum is 8 in-lane values, two cross-lane steps and one exchange between the partner waves through LDS.
template <bool QOUT>
__global__ __launch_bounds__(512, 1) void value_proj_kernel(const ValueProjArgs g)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int pq = lane & 7, cs = lane >> 3;        // loader role: pixels 4pq .. 4pq+3, channels 32 wave + 4cs .. +3
    const long m0 = (long)blockIdx.x * g.pixels_per_wg;
    if (m0 >= g.M)
        return;
    const long mend = m0 + g.pixels_per_wg < g.M ? m0 + g.pixels_per_wg : g.M;
    const int nstages = (int)((mend - m0 + VP_ROWS - 1) / VP_ROWS);
    float *smax = reinterpret_cast<float *>(lds + 2 * VP_BUF);          // [8 waves][32 pixels]

    auto load = [&](int st, vp_f4 *xv) {
        const long gp = m0 + (long)st * VP_ROWS;                          // first pixel of the stage (HW % 32 == 0: one frame)
        const long f = gp / g.HW;
        const int p = (int)(gp - f * g.HW) + 4 * pq;
        const float *src = g.x + (f * 256 + 32 * wave + 4 * cs) * (long)g.HW + p;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xv[i] = *reinterpret_cast<const vp_f4 *>(src + (long)i * g.HW);
    };
    vp_f4 xv[4];
    load(0, xv);

    // ---- this wave's weights: fragment (tile t, K step ks) = W rows 32 wave + 16t + li, chunk lk of the hi / lo half
    vp_h8 wh[2][8], wl[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const char *wp = g.w + (size_t)(32 * wave + 16 * t + li) * 1024 + lk * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            wh[t][ks] = *reinterpret_cast<const vp_h8 *>(wp + ks * 128);
            wl[t][ks] = *reinterpret_cast<const vp_h8 *>(wp + ks * 128 + 64);
        }
    }
    vp_f4 bias4[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
        bias4[t] = (g.bias && !g.add) ? *reinterpret_cast<const vp_f4 *>(g.bias + 32 * wave + 16 * t + 4 * lk) : (vp_f4){0.f, 0.f, 0.f, 0.f};

    // Everything issued so far (the 32 weight fragments, stage 0's pixels) has to have LANDED before the stage loop is entered,
    // and the compiler has to know it: hipcc places the vmcnt waits for the weight registers at their first uses INSIDE the loop
    // (vmcnt(30) ... vmcnt(0) between the MFMAs), and from the second stage on those same waits drain the loads of the NEXT
    // stage that were issued just above them -- the prefetch ran inside the MFMA phase instead of under it (round-4 ISA reading:
    // 63 % of the wave cycles parked in s_waitcnt).  simm16 = vmcnt(0), expcnt / lgkmcnt untouched.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int st = 0; st < nstages; ++st) {
        char *B = lds + (st & 1) * VP_BUF;
        float *salpha = reinterpret_cast<float *>(B + VP_ROWS * 1024);
        // ---- per-pixel maximum over the 256 channels: this lane's 4, the wave's 32 (lanes that differ in cs), the 8 waves (LDS)
        vp_f4 mx;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            mx[j] = fmaxf(fmaxf(fabsf(xv[0][j]), fabsf(xv[1][j])), fmaxf(fabsf(xv[2][j]), fabsf(xv[3][j])));
#pragma unroll
        for (int m = 8; m <= 32; m <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], m, 64));
        if (cs == 0)
            *reinterpret_cast<vp_f4 *>(smax + wave * VP_ROWS + 4 * pq) = mx;
        __syncthreads();                                  // (A) partial maxima visible; everyone is past the MFMAs of stage st-1
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            const vp_f4 o = *reinterpret_cast<const vp_f4 *>(smax + ww * VP_ROWS + 4 * pq);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                mx[j] = fmaxf(mx[j], o[j]);
        }
        // scale 2^(13 - e) with e = floor(log2(max)): exponent arithmetic only (max = 0 / denormal / inf: clamped exponents)
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int eb = (int)((__float_as_uint(mx[j]) >> 23) & 255u);
            eb = eb < 14 ? 14 : (eb > 254 ? 254 : eb);
            sc[j] = __uint_as_float((unsigned)(267 - eb) << 23);
            if (wave == 0 && cs == 0)
                salpha[4 * pq + j] = g.w_alpha * __uint_as_float((unsigned)(eb - 13) << 23);
        }
        // hi / lo images of the stage: pixel r = 4pq + j -> row r, K step = this wave, the lane's 4 channels = half a chunk
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * pq + j;
            vp_h4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16 h, l;
                rac_split_f16(xv[i][j] * sc[j], h, l);
                hi[i] = h;
                lo[i] = l;
            }
            char *rowp = B + r * 1024 + (cs & 1) * 8;
            *reinterpret_cast<vp_h4 *>(rowp + (((8 * wave + (cs >> 1)) ^ (r & 15)) * 16)) = hi;
            *reinterpret_cast<vp_h4 *>(rowp + (((8 * wave + 4 + (cs >> 1)) ^ (r & 15)) * 16)) = lo;
        }
        __syncthreads();                                  // (B) the stage's X image is complete
        // this stage's additive term first, then the next stage's pixels: both in flight under the MFMAs, and the epilogue's
        // wait for the (older) additive term leaves the pixel loads outstanding (loaded inside the epilogue, each of the four
        // loads was followed by a vmcnt(0) that drained the prefetch as well)
        const long gp = m0 + (long)st * VP_ROWS;
        const int p0 = (int)(gp % g.HW);
        vp_f4 addv[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const long row = min(gp + 16 * j + li, mend - 1) - gp;          // (rows past the end re-read the last row; not stored)
                addv[j][t] = g.add ? *reinterpret_cast<const vp_f4 *>(g.add + (size_t)(p0 + row) * 256 + 32 * wave + 16 * t + 4 * lk) : bias4[t];
            }
        // (unconditional -- past the end the last stage is fetched again and never used: a branch around these loads would make
        //  the number of loads in flight path-dependent, and the epilogue's wait for the additive term would fall back to vmcnt(0))
        load(st + 1 < nstages ? st + 1 : st, xv);         // in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);

        vp_f4 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[t][j] = (vp_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char *rowp = B + (16 * j + li) * 1024;
                const vp_h8 xh = *reinterpret_cast<const vp_h8 *>(rowp + (((8 * ks + lk) ^ li) * 16));
                const vp_h8 xl = *reinterpret_cast<const vp_h8 *>(rowp + (((8 * ks + 4 + lk) ^ li) * 16));
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t][ks], xh, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][ks], xl, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][ks], xh, acc[t][j], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: C/D layout col = li (pixel), rows 4 lk + r = four consecutive features: one 16-byte store
        if (QOUT) {
            unsigned *pmax = reinterpret_cast<unsigned *>(lds + VP_LDS);          // [8 waves][32 pixels]
            vp_f4 v[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a = salpha[16 * j + li];
                unsigned bm = 0u;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    v[j][t] = __builtin_elementwise_fma(acc[t][j], (vp_f4){a, a, a, a}, addv[j][t]);
                    bm = max(bm, rac_absbits4(v[j][t][0], v[j][t][1], v[j][t][2], v[j][t][3]));
                }
                bm = max(bm, (unsigned)__shfl_xor((int)bm, 16, 64));
                bm = max(bm, (unsigned)__shfl_xor((int)bm, 32, 64));
                if (lk == 0)
                    pmax[wave * VP_ROWS + 16 * j + li] = bm;
            }
            __syncthreads();                              // (C) the partner wave's half-block maxima are visible
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 16 * j + li;
                const unsigned bm = max(pmax[wave * VP_ROWS + r], pmax[(wave ^ 1) * VP_ROWS + r]);
                float up, dn;
                rac_q16_factors(bm, up, dn);
                if (gp + r < mend) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        *reinterpret_cast<uint2 *>(g.q + (size_t)(gp + r) * 256 + 32 * wave + 16 * t + 4 * lk) =
                            rac_q16x4(v[j][t][0], v[j][t][1], v[j][t][2], v[j][t][3], up);
                    if (lk == 0 && (wave & 1) == 0)
                        g.qscale[(size_t)(gp + r) * 4 + (wave >> 1)] = dn;
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 16 * j + li;
            const float a = salpha[r];
            if (gp + r < mend) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int n = 32 * wave + 16 * t + 4 * lk;
                    *reinterpret_cast<vp_f4 *>(g.out + (size_t)(gp + r) * 256 + n) = __builtin_elementwise_fma(acc[t][j], (vp_f4){a, a, a, a}, addv[j][t]);
                }
            }
        }
    }
}

static int vp_launch(const float *x, const void *w_image, float w_alpha, const float *add, const float *bias, float *out, void *q,
                     float *qscale, int frames, int channels, int HW, int features, void *stream, const char *what)
{
    RAC_CHECK_ARG(channels == 256 && features == 256, "%s: built for 256 -> 256 (got %d -> %d)", what, channels, features);
    RAC_CHECK_ARG(frames >= 0 && HW >= VP_ROWS && HW % VP_ROWS == 0, "%s: frames=%d H*W=%d (H*W must be a multiple of %d)", what,
                  frames, HW, VP_ROWS);
    if (frames == 0)
        return 0;
    RAC_CHECK_ARG(x && w_image && (out || (q && qscale)), "%s: null pointer", what);
    RAC_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(add) |
                    reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(q)) & 15) == 0, "%s: pointers must be 16-byte aligned", what);
    ValueProjArgs a;
    a.x = x; a.w = reinterpret_cast<const char *>(w_image); a.add = add; a.bias = bias; a.out = out; a.w_alpha = w_alpha;
    a.q = reinterpret_cast<short *>(q); a.qscale = qscale;
    a.HW = HW; a.M = (long)frames * HW;
    // one workgroup per CU (its weights fill the register file): cut the pixels into about 256 chunks of whole stages
    long ppw = (a.M + 255) / 256;
    ppw = (ppw + VP_ROWS - 1) / VP_ROWS * VP_ROWS;
    a.pixels_per_wg = (int)ppw;
    const unsigned grid = (unsigned)((a.M + ppw - 1) / ppw);
    if (q) {
        if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_VALUE_PROJ_Q16, reinterpret_cast<const void *>(value_proj_kernel<true>), (int)(VP_LDS_Q16)))
            return rc_attr;
        hipLaunchKernelGGL(value_proj_kernel<true>, dim3(grid), dim3(512), VP_LDS_Q16, (hipStream_t)stream, a);
    } else {
        if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_VALUE_PROJ, reinterpret_cast<const void *>(value_proj_kernel<false>), (int)(VP_LDS)))
            return rc_attr;
        hipLaunchKernelGGL(value_proj_kernel<false>, dim3(grid), dim3(512), VP_LDS, (hipStream_t)stream, a);
    }
    return rac_launch_status(what);
}

extern "C" int rac_value_proj_fwd(const float *x, const void *w_image, float w_alpha, const float *add, const float *bias, float *out,
                                  int frames, int channels, int HW, int features, void *stream)
{
    return vp_launch(x, w_image, w_alpha, add, bias, out, nullptr, nullptr, frames, channels, HW, features, stream, "rac_value_proj_fwd");
}

extern "C" int rac_value_proj_q16_fwd(const float *x, const void *w_image, float w_alpha, const float *add, const float *bias, void *q,
                                      float *scale, int frames, int channels, int HW, int features, void *stream)
{
    RAC_CHECK_ARG(frames == 0 || (q && scale), "rac_value_proj_q16_fwd: null pointer");
    return vp_launch(x, w_image, w_alpha, add, bias, nullptr, q, scale, frames, channels, HW, features, stream, "rac_value_proj_q16_fwd");
}
