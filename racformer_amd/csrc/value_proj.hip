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
// generator_ws_kernel) and walks `pixels_per_wg` pixels in stages of 32.  Per stage, wave w also takes in channels
// 32w .. 32w+31 of the 32 pixels (lane = pixel quad x channel quad: 4 x float4), the per-pixel maxima are combined across
// the waves through LDS, and the scaled hi / lo values go to LDS in the fragment layout of gemm_split.hip (1 KB per pixel,
// 16-byte chunk c at slot c ^ (pixel & 15)).
// Round 5: the pixels come through an LDS-DMA ring of raw fp32 stages (global_load_lds_dwordx4, VP_RING stages of 32 KB, each wave
// its own 4 KB of a stage: no barrier between a wave's DMA and its own reads), issued VP_RING stages ahead.  Rounds 2-4 loaded
// the next stage into registers under the MFMAs: with 252 VGPRs there was room for ONE stage, i.e. 32 KB in flight per CU, and at
// ~2.5 us of loaded HBM latency that is 13 GB/s per CU = 3.3 TB/s chip-wide -- the kernel's 99-113 us.  (Round 4 had blamed the two
// barriers per stage.)
#include "rac_common.h"

typedef _Float16 vp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 vp_h4 __attribute__((ext_vector_type(4)));
typedef float vp_f4 __attribute__((ext_vector_type(4)));

#define VP_ROWS 32
#ifndef VP_RING
#define VP_RING 2                                /* raw stages in the LDS-DMA ring = stages in flight (the stage loop is unrolled by it) */
#endif
#define VP_BUF (VP_ROWS * 1024 + VP_ROWS * 4)   /* X image of a stage + its per-row output scales */
#define VP_RAW (VP_ROWS * 256 * 4)              /* one raw stage: [256 channels][32 pixels] fp32 */
#define VP_RAW0 (VP_BUF + 8 * VP_ROWS * 4)      /* X image (one buffer: barrier (A) of a stage comes after every wave's MFMAs of the one before), partial maxima, then the ring */
#define VP_LDS (VP_RAW0 + VP_RING * VP_RAW)
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
    unsigned x_bytes;    // F * 256 * HW * 4
    long M;              // F * HW
    int pixels_per_wg;   // multiple of VP_ROWS
};

// QOUT: the result leaves in the int16 block storage of quant.hip (what rac_bev_sampling_multi_q16_fwd reads), bit for bit what
// rac_quant_i16_fwd makes of the fp32 output: a (pixel, head) block of 64 features is held by two waves (32 features each), so
// the block maximum is 8 in-lane values, two cross-lane steps and one exchange between the partner waves through LDS.
// HAS_ADD: the additive term is a per-pixel map (loaded per stage) / a per-feature bias in registers -- a template parameter: as
// a run-time choice it put a branch around each of the stage's four additive loads, the number of loads in flight became
// path-dependent and the epilogue's wait for them fell back to vmcnt(0), draining the pixel ring every stage.
// The body is a function of its own with the LDS regions as __restrict__ parameters -- the X image (+ maxima, scales), the two ring
// slots, the q16 maxima: inlining turns them into alias scopes, and that is what lets hipcc tell a ring slot's pending LDS-DMA from
// reads of the other slot or of the image.  Without scope information every LDS access waits for ALL LDS-DMA in flight
// (SIInsertWaitcnts), i.e. the stage that was prefetched had to land before the current one could even be read.
template <bool QOUT, bool HAS_ADD>
__device__ __forceinline__ void value_proj_body(const ValueProjArgs &g, char *__restrict__ lds, char *__restrict__ ring0,
                                                char *__restrict__ ring1, unsigned *__restrict__ pmax)
{
    static_assert(VP_RING == 2, "one __restrict__ parameter per ring slot");
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int pq = lane & 7, cs = lane >> 3;        // loader role: pixels 4pq .. 4pq+3, channels 32 wave + 4cs .. +3
    // (pixel indices are 32-bit: the launcher checks F * HW < 2^31; 64-bit divisions per stage cost ~100 scalar instructions each)
    const int Mi = (int)g.M;
    const int m0 = (int)blockIdx.x * g.pixels_per_wg;
    if (m0 >= Mi)
        return;
    const int mend = min(m0 + g.pixels_per_wg, Mi);
    const int nstages = (mend - m0 + VP_ROWS - 1) / VP_ROWS;
    float *smax = reinterpret_cast<float *>(lds + VP_BUF);              // [8 waves][32 pixels]

    // raw stage st -> ring slot: this wave's 32 channel rows (128 B each) as four 1 KB LDS-DMA pieces of 8 rows.  Piece position q
    // (0..7) receives channel row q ^ ((q >> 2) & 1): rows r and r + 4 -- which a quarter-wave of the readers below touches together --
    // land at different parities of the 128-byte row grid (different halves of the 64 banks).
    const int dq = lane >> 3, drow = dq ^ ((dq >> 2) & 1), dslot = lane & 7;
    // (buffer_load ... lds over a descriptor of the maps, not global_load_lds: hipcc books the global form as a FLAT access that may
    //  return out of order and turns every later vector-memory wait into vmcnt(0) -- the epilogue's wait for the additive term would
    //  drain the ring each stage; the buffer form gets counted waits)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.x), 0, g.x_bytes, 0x00020000);
    auto dma = [&](int st, int slot) {
        const int stc = st < nstages ? st : nstages - 1;                  // (past the end: the last stage again, never read)
        const int gp = m0 + stc * VP_ROWS;                                // first pixel of the stage (HW % 32 == 0: one frame)
        const int f = gp / g.HW;
        const int p = gp - f * g.HW + 4 * dslot;
        const unsigned voff = (unsigned)(((f * 256 + 32 * wave + drow) * g.HW + p) * 4);     // (the launcher checks the maps stay below 4 GiB)
        char *dst = (slot == 0 ? ring0 : ring1) + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(dst + i * 1024), 16,
                                                     voff + (unsigned)(8 * i * g.HW * 4), 0, 0, 0);
    };
    // reader role: channel rows 4cs .. 4cs+3 of the wave's 32, pixels 4pq .. 4pq+3
    int roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * cs + i, q = r & 7;
        roff[i] = wave * 4096 + (r >> 3) * 1024 + (q ^ ((q >> 2) & 1)) * 128 + pq * 16;
    }
#pragma unroll
    for (int s0 = 0; s0 < VP_RING; ++s0)
        dma(s0, s0);

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
        bias4[t] = (g.bias && !HAS_ADD) ? *reinterpret_cast<const vp_f4 *>(g.bias + 32 * wave + 16 * t + 4 * lk) : (vp_f4){0.f, 0.f, 0.f, 0.f};

    // Everything issued so far (the 32 weight fragments, stage 0's pixels) has to have LANDED before the stage loop is entered,
    // and the compiler has to know it: hipcc places the vmcnt waits for the weight registers at their first uses INSIDE the loop
    // (vmcnt(30) ... vmcnt(0) between the MFMAs), and from the second stage on those same waits drain the loads of the NEXT
    // stage that were issued just above them -- the prefetch ran inside the MFMA phase instead of under it (round-4 ISA reading:
    // 63 % of the wave cycles parked in s_waitcnt).  simm16 = vmcnt(0), expcnt / lgkmcnt untouched.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int st0 = 0; st0 < nstages; st0 += VP_RING)
#pragma unroll
    for (int slot = 0; slot < VP_RING; ++slot) {
        const int st = st0 + slot;
        if (st >= nstages)
            break;
        char *B = lds;
        float *salpha = reinterpret_cast<float *>(B + VP_ROWS * 1024);
        // this wave's rows of stage st have landed: everything issued behind that DMA may still be in flight -- per stage the
        // additive term (4 loads), the next DMA (4 pieces) and the stores (>= 4): VP_RING - 1 whole stages + the stores of the stage
        // that issued it.  (First trip: the weights' vmcnt(0) above covered it.)
        // (without an additive map the stage issues no such loads: 8 per stage)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((HAS_ADD ? 12 : 8) * (VP_RING - 1) + 4) : "memory");
        // The four reads of the raw stage are inline assembly: as ordinary LDS loads hipcc cannot tell them from the OTHER slot's pending
        // LDS-DMA (run-time lane offsets into one dynamic LDS array: SIInsertWaitcnts then waits for every LDS-DMA in flight, here
        // vmcnt(4), i.e. for the stage that has just been prefetched -- measured in the ISA with and without __restrict__ regions).
        // The counted wait above is what orders them behind their own stage's pieces.
        vp_f4 xv[4];
        {
            const unsigned rbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(slot == 0 ? ring0 : ring1);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3])
                         : "v"(rbase + roff[0]), "v"(rbase + roff[1]), "v"(rbase + roff[2]), "v"(rbase + roff[3]));
            // (no "memory" clobber: with one the block counts as a possible reader of the LDS-DMA's destination and gets the very wait it
            //  is there to avoid; volatile keeps it behind the counted wait above and in front of the barrier below)
        }
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
        const int gp = m0 + st * VP_ROWS;
        const int p0 = gp % g.HW;
        vp_f4 addv[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 16 * j + li;       // (every stage is whole: H*W, hence M and pixels_per_wg, are multiples of VP_ROWS)
                if constexpr (HAS_ADD)
                    addv[j][t] = *reinterpret_cast<const vp_f4 *>(g.add + (size_t)(p0 + row) * 256 + 32 * wave + 16 * t + 4 * lk);
                else
                    addv[j][t] = bias4[t];
            }
        // the ring slot this stage has just been read out of (into registers, before barrier (A)) takes stage st + VP_RING
        // (unconditional -- past the end the last stage is fetched again and never read: the counted waits assume it)
        dma(st + VP_RING, slot);
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
            // pmax [8 waves][32 pixels]
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
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    *reinterpret_cast<uint2 *>(g.q + (size_t)(gp + r) * 256 + 32 * wave + 16 * t + 4 * lk) =
                        rac_q16x4(v[j][t][0], v[j][t][1], v[j][t][2], v[j][t][3], up);
                if (lk == 0 && (wave & 1) == 0)
                    g.qscale[(size_t)(gp + r) * 4 + (wave >> 1)] = dn;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 16 * j + li;
            const float a = salpha[r];
            // (no range check around the stores: every stage is whole, and a branch around a store makes the number of vector-memory
            //  operations in flight path-dependent -- hipcc then sizes every later counted wait for the path WITHOUT the stores, which on
            //  the real path waits for the ring's pieces issued a microsecond ago)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = 32 * wave + 16 * t + 4 * lk;
                *reinterpret_cast<vp_f4 *>(g.out + (size_t)(gp + r) * 256 + n) = __builtin_elementwise_fma(acc[t][j], (vp_f4){a, a, a, a}, addv[j][t]);
            }
        }
    }
}

template <bool QOUT, bool HAS_ADD>
__global__ __launch_bounds__(512, 1) void value_proj_kernel(const ValueProjArgs g)
{
    extern __shared__ char vp_lds[];
    value_proj_body<QOUT, HAS_ADD>(g, vp_lds, vp_lds + VP_RAW0, vp_lds + VP_RAW0 + VP_RAW, reinterpret_cast<unsigned *>(vp_lds + VP_LDS));
}

static int vp_launch(const float *x, const void *w_image, float w_alpha, const float *add, const float *bias, float *out, void *q,
                     float *qscale, int frames, int channels, int HW, int features, void *stream, const char *what)
{
    RAC_CHECK_ARG(channels == 256 && features == 256, "%s: built for 256 -> 256 (got %d -> %d)", what, channels, features);
    RAC_CHECK_ARG(frames >= 0 && HW >= VP_ROWS && HW % VP_ROWS == 0, "%s: frames=%d H*W=%d (H*W must be a multiple of %d)", what,
                  frames, HW, VP_ROWS);
    RAC_CHECK_ARG((long)frames * 256 * HW * 4 < (1l << 32), "%s: the maps (%ld bytes) exceed the kernel's 32-bit buffer offsets", what,
                  (long)frames * 256 * HW * 4);
    if (frames == 0)
        return 0;
    RAC_CHECK_ARG(x && w_image && (out || (q && qscale)), "%s: null pointer", what);
    RAC_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(add) |
                    reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(q)) & 15) == 0, "%s: pointers must be 16-byte aligned", what);
    ValueProjArgs a;
    a.x = x; a.w = reinterpret_cast<const char *>(w_image); a.add = add; a.bias = bias; a.out = out; a.w_alpha = w_alpha;
    a.q = reinterpret_cast<short *>(q); a.qscale = qscale;
    a.HW = HW; a.M = (long)frames * HW;
    a.x_bytes = (unsigned)((long)frames * 256 * HW * 4);
    // one workgroup per CU (its weights fill the register file): cut the pixels into about 256 chunks of whole stages
    long ppw = (a.M + 255) / 256;
    ppw = (ppw + VP_ROWS - 1) / VP_ROWS * VP_ROWS;
    a.pixels_per_wg = (int)ppw;
    const unsigned grid = (unsigned)((a.M + ppw - 1) / ppw);
#define VP_GO(Q_, A_, ATTR_, LDS_)                                                                                                 \
    do {                                                                                                                           \
        if (const int rc_attr = rac_set_dynamic_lds_once(ATTR_, reinterpret_cast<const void *>(value_proj_kernel<Q_, A_>), (int)(LDS_))) \
            return rc_attr;                                                                                                        \
        hipLaunchKernelGGL((value_proj_kernel<Q_, A_>), dim3(grid), dim3(512), LDS_, (hipStream_t)stream, a);                        \
    } while (0)
    if (q && add) VP_GO(true, true, RAC_ATTR_VALUE_PROJ_Q16, VP_LDS_Q16);
    else if (q) VP_GO(true, false, RAC_ATTR_VALUE_PROJ_Q16_BIAS, VP_LDS_Q16);
    else if (add) VP_GO(false, true, RAC_ATTR_VALUE_PROJ, VP_LDS);
    else VP_GO(false, false, RAC_ATTR_VALUE_PROJ_BIAS, VP_LDS);
#undef VP_GO
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
