// conv3x3.hip -- the temporal-fusion convolution of RadarBEVTemporalEncoder (3x3, stride 1, pad 1, 320 -> 256
// channels on 8 x 128 x 128 BEV maps: 193 of the encoder's 220 GFLOP; models/racformer_transformer.py:645-656) as an
// implicit GEMM on the f16 matrix cores at fp32-GEMM accuracy (gfx950).
//
// Arithmetic: every activation and weight is split as v * 2^e = hi + lo (two f16, 22 significant bits) and the
// three leading products hi*hi + hi*lo + lo*hi are accumulated in fp32 by v_mfma_f32_16x16x32_f16 -- the dropped
// lo*lo term is 2^-22 relative, the result matches an fp32 convolution to fp32 rounding.  The power of two 2^e
// of the activations is derived on the device from their max |value| (rac_absmax_fwd), so no input magnitude can
// overflow f16; the weights' 2^e is chosen by the host when it packs them.  Three f16 MFMA products cost 3/16 of
// the f32-input MFMA time of the same GEMM.
//
// Layouts (all built for this kernel):
//   activations  xs [N][H+2][W+2][Cin/32][2][32] f16  zero border (the conv's padding), per pixel and 32-channel
//                chunk 64 B of hi then 64 B of lo: one K-step's A row is one 128-byte line
//   weights      ws [9 taps][Cin/32][Cout=256][2][32] f16: one K-step's B tile is 32 KB contiguous
//   output       out [N][H][W][256] f32 (channel-last: what value_proj's GEMM reads as its A operand)
// Workgroup = 512 threads = 8 waves (2 along pixels x 4 along channels), tile = 256 consecutive pixels of one
// image x 256 output channels, K loop over 9 taps x Cin/32 chunks.  Per K-step 64 KB (A 32 KB + B 32 KB) go
// global -> registers -> LDS (two LDS stages = 128 KB; tile k+1 is written to LDS at the start of step k and the
// registers re-used for the loads of tile k+2, which land under step k's MFMAs; one barrier per step), fragments are 16-byte LDS reads made conflict-free by an
// XOR swizzle of the 16-byte slots (slot ^ ((row >> 1) & 7)), 96 MFMAs per wave and step.
#include "rac_common.h"
#include <string.h>

typedef _Float16 cv_h8 __attribute__((ext_vector_type(8)));
typedef float cv_f4 __attribute__((ext_vector_type(4)));

#ifndef CV_TAP_MAJOR
#define CV_TAP_MAJOR 0
#endif
#ifndef CV_SPREAD_STAGING
#define CV_SPREAD_STAGING 1   /* 0: all staging traffic at the top of the step (rounds 2-3; A/B builds) */
#endif
#define CV_TM 256
#define CV_COUT 256
#define CV_STAGE_U4 4096 /* uint4 per LDS stage: A 2048 + B 2048 */

// power of two that brings amax into [2^13, 2^14): rac_act_scale (rac_common.h; shared with conv_direct.hip)
#define cv_act_scale rac_act_scale

// ------------------------------------------------------------------------------------------------ absmax
// (a one-thread launch rather than a memset: inside a captured HIP graph the step then consists of kernel nodes only, whose
//  order along the stream is the order of execution)
__global__ void absmax_init_kernel(unsigned *__restrict__ out, unsigned floor_bits)
{
    *out = floor_bits;
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ src, long n, unsigned *__restrict__ out)
{
    float m = 0.f;
    const long n4 = n >> 2, stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {   // eight independent 16-byte loads in flight per lane
        rac_f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = rac_ld4(src + (i + u * stride) * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
    }
    for (; i < n4; i += stride) {
        const rac_f4 v = rac_ld4(src + i * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
        m = fmaxf(m, fabsf(src[n4 * 4 + threadIdx.x]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        // non-negative floats order like their bit patterns.  The word only ever grows, so a block whose maximum does not exceed
        // what it reads there has nothing to add: after the first few blocks almost none issues the atomic (2048 atomics on one
        // address serialise in L2)
        const unsigned bits = __float_as_uint(m);
        if (bits > __atomic_load_n(out, __ATOMIC_RELAXED))
            atomicMax(out, bits);
    }
}

// ------------------------------------------------------------------------------------------------ pack
// [N][C][H][W] f32 -> channel chunks chunk0.. of xs (interior pixels only; the border stays zero).
// Optional per-channel bias, and frames in groups of T of which only the first Tv exist in src (group g, frame t < Tv: src frame
// g*Tv + t, plus the bias; frames t >= Tv: the bias alone) -- the hidden half of the temporal-fusion input, whose frames past the
// ConvGRU's live ones are exactly the last convolution's bias.
__global__ __launch_bounds__(256) void conv_pack_kernel(const float *__restrict__ src, const float *__restrict__ bias,
                                                        const float *__restrict__ amax, _Float16 *__restrict__ dst, int C, int H, int W,
                                                        int chunks_total, int chunk0, int T, int Tv)
{
    // one block = 32 channels x one image row x up to 128 columns (wider rows -- the FPN's 176 -- take several blocks)
    __shared__ float tile[32][129];
    const int cchunks = C >> 5, wpieces = (W + 127) >> 7;
    const int wp = blockIdx.x % wpieces;
    const int c32 = (blockIdx.x / wpieces) % cchunks;
    const int h = (blockIdx.x / (wpieces * cchunks)) % H;
    const int n = blockIdx.x / (wpieces * cchunks * H);
    const int tid = threadIdx.x;
    const float scale = cv_act_scale(*amax);
    const int w0 = wp << 7, wlen = min(128, W - w0);
    const int grp = n / T, t = n - grp * T;
    const bool live = t < Tv;
    const size_t ns = (size_t)grp * Tv + t;             // source frame
    if ((W & 3) == 0) {
        const int w4n = wlen >> 2;
        for (int i = tid; i < 32 * w4n; i += 256) {
            const int ch = i / w4n, w4 = i - ch * w4n;
            rac_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (live)
                v = rac_ld4(src + ((ns * C + c32 * 32 + ch) * H + h) * W + w0 + w4 * 4);
            if (bias) {
                const float bch = bias[c32 * 32 + ch];
                v.x += bch; v.y += bch; v.z += bch; v.w += bch;
            }
            tile[ch][w4 * 4 + 0] = v.x;
            tile[ch][w4 * 4 + 1] = v.y;
            tile[ch][w4 * 4 + 2] = v.z;
            tile[ch][w4 * 4 + 3] = v.w;
        }
    } else {
        for (int i = tid; i < 32 * wlen; i += 256) {     // rows are not 16-byte aligned: scalar loads (coarse FPN levels only)
            const int ch = i / wlen, w = i - ch * wlen;
            float v = live ? src[((ns * C + c32 * 32 + ch) * H + h) * W + w0 + w] : 0.f;
            if (bias)
                v += bias[c32 * 32 + ch];
            tile[ch][w] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < wlen * 8; i += 256) {
        const int w = i >> 3, s = i & 7, c0 = (s & 3) * 8;
        cv_h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = tile[c0 + j][w] * scale;
            const _Float16 hi = (_Float16)v;
            o[j] = s < 4 ? hi : (_Float16)(v - (float)hi);
        }
        const size_t pix = ((size_t)n * (H + 2) + h + 1) * (W + 2) + w0 + w + 1;
        *reinterpret_cast<cv_h8 *>(dst + (pix * chunks_total + chunk0 + c32) * 64 + s * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------ conv
RAC_CLOCK_DECL(conv3x3)
RAC_CLOCK_READER(conv3x3)
struct ConvArgs {
    const uint4 *xs;
    const uint4 *ws;
    const float *bias;        // [256] per output channel, or
    const float *pixel_bias;  // [H*W][256] per pixel of an image and output channel (then bias is unused)
    const float *amax;
    float *out;
    int N, H, W, chunks;
    float w_alpha;
    int cams;                 // GROUPED output only: images per (batch, frame)
    short *q;                 // Q16 output only: [N][H*W][256] int16 mantissas and
    float *qscale;            //   [N][H*W][4] one power-of-two scale per (pixel, 64-channel block); out is unused
    // rac_conv3x3_temporal_fwd: images come in groups of `fpg` frames (0: no grouping) of which the first `live` carry all chunks; the
    // others run only `chunks_dead` chunks per tap and add `pixel_bias_dead` -- their remaining channels are a per-channel constant
    // whose contribution through the zero padding the caller folded into that map
    const float *pixel_bias_dead;
    int fpg, live, chunks_dead;
};
enum { CV_OUT_NHWC = 0, CV_OUT_GROUPED = 1, CV_OUT_Q16 = 2 };
__device__ __forceinline__ cv_f4 cv_fma4(cv_f4 a, float s, cv_f4 b)
{
    return __builtin_elementwise_fma(a, (cv_f4){s, s, s, s}, b);
}

// MODE = CV_OUT_NHWC: out [N][H][W][256] channel-last.  CV_OUT_GROUPED: the decoder's sampling layout of a feature-pyramid
// level, [N / cams * 4][cams][H][W][64] -- image n = (b*T + t) * cams + cam, output channel co = g * 64 + c lands in slot
// (b*T + t) * 4 + g (models/racformer_transformer.py:112-124: what the reference builds with a reshape / permute copy of
// the FPN's output; here the FPN's last convolution writes it).  CV_OUT_Q16: the channel-last result in the int16 block
// storage of quant.hip (the BEV value stream as rac_bev_sampling_multi_q16_fwd reads it), quantised in the epilogue: bit for
// bit what rac_quant_i16_fwd makes of the CV_OUT_NHWC output, without the 134 MB fp32 stream in between.
template <int MODE>
__global__ __launch_bounds__(512, 1) void conv3x3_f16x3_kernel(const ConvArgs a)
{
    constexpr bool GROUPED = MODE == CV_OUT_GROUPED;
    extern __shared__ uint4 lds4[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int H = a.H, W = a.W, Wp = W + 2, chunks = a.chunks;
    const int HW = H * W;
    const int tiles_per_img = (HW + CV_TM - 1) / CV_TM;    // the last tile of an image may be ragged: its rows past the image
    const int n = blockIdx.x / tiles_per_img, tile = blockIdx.x - n * tiles_per_img;   // re-read the last pixel and are not stored
    // (round 5) frames past a group's live ones: fewer chunks per tap, another per-pixel map; the image and weight strides stay `chunks`
    const bool dead = a.fpg > 0 && (n % a.fpg) >= a.live;
    const int KS = 9 * (dead ? a.chunks_dead : chunks);
    const size_t pix_stride = (size_t)chunks * 8;   // uint4 per pixel

    // staging role: 16-byte slot (tid & 7) of rows (tid >> 3) + 64 j of both the A and the B tile
    auto row_base = [&](int j) -> size_t {
        const int r = (tid >> 3) + 64 * j;
        const int gp = min(tile * CV_TM + r, HW - 1), h = gp / W, w = gp - h * W;
        return (((size_t)n * (H + 2) + h) * Wp + w) * pix_stride + (tid & 7);
    };
    auto row_slot = [&](int j) -> int {
        const int r = (tid >> 3) + 64 * j;
        return r * 8 + ((tid & 7) ^ ((r >> 1) & 7));
    };
    const size_t a_base0 = row_base(0), a_base1 = row_base(1), a_base2 = row_base(2), a_base3 = row_base(3);
    const int st0 = row_slot(0), st1 = row_slot(1), st2 = row_slot(2), st3 = row_slot(3);
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    // K order: chunk-major, the 9 taps of a 32-channel chunk back to back -- they re-read the same 4 image rows of that chunk
    // (66 KB per workgroup, 2 MB per XCD: L2 hits), where tap-major had 10 chunks = 320 KB per workgroup between two reads of a
    // line (10 MB per XCD > the 4 MB L2: every tap came from the Infinity Cache -- PMC 1.62 GB fetched per launch for a 173 MB
    // image).  CV_TAP_MAJOR = 1: the old order (A/B builds).
#define CV_GLOAD(ks_)                                                                          \
    do {                                                                                       \
        const int chunk_ = CV_TAP_MAJOR ? (ks_) % chunks : (ks_) / 9;                          \
        const int tap_ = CV_TAP_MAJOR ? (ks_) / chunks : (ks_) - chunk_ * 9;                   \
        const int dy_ = tap_ / 3, dx_ = tap_ - dy_ * 3;                                        \
        const size_t off_ = ((size_t)dy_ * Wp + dx_) * pix_stride + (size_t)chunk_ * 8;        \
        const uint4 *wsrc_ = a.ws + (size_t)(tap_ * chunks + chunk_) * 2048 + tid;             \
        ra0 = a.xs[a_base0 + off_]; ra1 = a.xs[a_base1 + off_];                            \
        ra2 = a.xs[a_base2 + off_]; ra3 = a.xs[a_base3 + off_];                            \
        rb0 = wsrc_[0]; rb1 = wsrc_[512]; rb2 = wsrc_[1024]; rb3 = wsrc_[1536];        \
    } while (0)
#define CV_LSTORE(buf_)                                                                        \
    do {                                                                                       \
        uint4 *A_ = lds4 + (buf_) * CV_STAGE_U4, *B_ = A_ + 2048;                              \
        A_[st0] = ra0; A_[st1] = ra1; A_[st2] = ra2; A_[st3] = ra3;                    \
        B_[st0] = rb0; B_[st1] = rb1; B_[st2] = rb2; B_[st3] = rb3;                    \
    } while (0)

    cv_f4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
            acc[m][nn] = (cv_f4){0.f, 0.f, 0.f, 0.f};

    // fragment slots: row = base + li, hi slot lk, lo slot 4 + lk, both XOR-swizzled with (row >> 1) & 7
    int bidx_h[4], bidx_l[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
        const int row = 64 * wn + 16 * nn + li, f = (row >> 1) & 7;
        bidx_h[nn] = 2048 + row * 8 + (lk ^ f);
        bidx_l[nn] = 2048 + row * 8 + ((4 + lk) ^ f);
    }
    const int arow0 = 128 * wm + li;

    // Pipeline: the registers always hold the tile after the one being multiplied.  At the top of step ks they (tile ks+1,
    // loaded during step ks-1) are written to the other LDS stage -- free since the barrier that ended step ks-1 -- and
    // immediately re-used for the loads of tile ks+2, which then have the whole MFMA phase to land; the LDS writes sit at
    // the start of a step, not between the last MFMA and the barrier.
    CV_GLOAD(0);
    CV_LSTORE(0);
    CV_GLOAD(KS > 1 ? 1 : 0);
    __syncthreads();
    RAC_CLOCK_BEGIN();
#if CV_SPREAD_STAGING
    // Round 4: the step's staging traffic -- eight LDS writes of tile ks+1 and the eight global loads of tile ks+2 that re-use
    // their registers -- is dealt out over the eight MFMA groups of the step, one (write, load) pair per group, and the A
    // fragments of group m+1 are read under the MFMAs of group m.  Before, all 64 ds_write_b128 of the workgroup hit the LDS
    // at the top of the step (830 cycles of write path, the first fragment reads queued behind them): the matrix pipe sat
    // idle for about a quarter of every step (PMC: 54 % MFMA-busy, 43 % of the wave cycles parked).
    for (int ks = 0; ks < KS; ++ks) {
        const int kn = ks + 2 < KS ? ks + 2 : KS - 1;   // (past the end the last tile is re-fetched / re-written: unconditional code)
        const int chunk_ = CV_TAP_MAJOR ? kn % chunks : kn / 9;
        const int tap_ = CV_TAP_MAJOR ? kn / chunks : kn - chunk_ * 9;
        const int dy_ = tap_ / 3, dx_ = tap_ - dy_ * 3;
        const size_t off_ = ((size_t)dy_ * Wp + dx_) * pix_stride + (size_t)chunk_ * 8;
        const uint4 *wsrc_ = a.ws + (size_t)(tap_ * chunks + chunk_) * 2048 + tid;
        uint4 *DA = lds4 + ((ks + 1) & 1) * CV_STAGE_U4, *DB = DA + 2048;
        const cv_h8 *S = reinterpret_cast<const cv_h8 *>(lds4 + (ks & 1) * CV_STAGE_U4);
        cv_h8 bh[4], bl[4], ah[2], al[2];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            bh[nn] = S[bidx_h[nn]];
            bl[nn] = S[bidx_l[nn]];
        }
        {
            const int row = arow0, f = (row >> 1) & 7;
            ah[0] = S[row * 8 + (lk ^ f)];
            al[0] = S[row * 8 + ((4 + lk) ^ f)];
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m + 1 < 8) {
                const int row = arow0 + 16 * (m + 1), f = (row >> 1) & 7;
                ah[(m + 1) & 1] = S[row * 8 + (lk ^ f)];
                al[(m + 1) & 1] = S[row * 8 + ((4 + lk) ^ f)];
            }
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(CV_ABLATE)
            // diagnostic builds only (wrong results, timing ablations: DESIGN 3.6): 1 = no LDS writes, 2 = no global loads, 4 = no step barrier
#define CV_ST(dst_, src_) do { if (!(CV_ABLATE & 1)) dst_ = src_; } while (0)
#define CV_LD(dst_, expr_) do { if (!(CV_ABLATE & 2)) dst_ = expr_; } while (0)
#else
#define CV_ST(dst_, src_) dst_ = src_
#define CV_LD(dst_, expr_) dst_ = expr_
#endif
            switch (m) {     // staging piece m: tile ks+1 out of its register into the other LDS stage, tile ks+2 into the register
            case 0: CV_ST(DA[st0], ra0); CV_LD(ra0, a.xs[a_base0 + off_]); break;
            case 1: CV_ST(DA[st1], ra1); CV_LD(ra1, a.xs[a_base1 + off_]); break;
            case 2: CV_ST(DA[st2], ra2); CV_LD(ra2, a.xs[a_base2 + off_]); break;
            case 3: CV_ST(DA[st3], ra3); CV_LD(ra3, a.xs[a_base3 + off_]); break;
            case 4: CV_ST(DB[st0], rb0); CV_LD(rb0, wsrc_[0]); break;
            case 5: CV_ST(DB[st1], rb1); CV_LD(rb1, wsrc_[512]); break;
            case 6: CV_ST(DB[st2], rb2); CV_LD(rb2, wsrc_[1024]); break;
            default: CV_ST(DB[st3], rb3); CV_LD(rb3, wsrc_[1536]); break;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[nn], al[m & 1], acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[nn], ah[m & 1], acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[nn], ah[m & 1], acc[m][nn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(CV_ABLATE) && (CV_ABLATE & 4)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        __syncthreads();
#endif
    }
#else
    for (int ks = 0; ks < KS; ++ks) {
        // (past the end the last tile is re-fetched / re-written into the unused stage: unconditional code keeps the
        //  staging registers out of scratch)
        CV_LSTORE((ks + 1) & 1);
        const int kn = ks + 2 < KS ? ks + 2 : KS - 1;
        CV_GLOAD(kn);
        // (nothing may move across this line: without it hipcc re-uses the eight staging registers for A fragments during the MFMA
        //  phase and sinks the loads of tile ks+2 to the END of the step)
        __builtin_amdgcn_sched_barrier(0);
        const cv_h8 *S = reinterpret_cast<const cv_h8 *>(lds4 + (ks & 1) * CV_STAGE_U4);
        cv_h8 bh[4], bl[4];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            bh[nn] = S[bidx_h[nn]];
            bl[nn] = S[bidx_l[nn]];
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int row = arow0 + 16 * m, f = (row >> 1) & 7;
            const cv_h8 ah = S[row * 8 + (lk ^ f)];
            const cv_h8 al = S[row * 8 + ((4 + lk) ^ f)];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[nn], al, acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[nn], ah, acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[nn], ah, acc[m][nn], 0, 0, 0);
        }
        __syncthreads();
    }
#endif

    RAC_CLOCK_END(conv3x3, blockIdx.x);
    // epilogue: undo the two power-of-two scalings, add the bias (per channel, or per pixel and channel), channel-last store.
    // The weights are the MFMA's A operand, so a 16x16 accumulator tile has its PIXEL on the lane (column li) and four
    // consecutive output CHANNELS (rows 4 lk + r) in the lane's registers: one 16-byte store (and one 16-byte bias load) per
    // tile instead of four 4-byte ones (round 4; the predicated 4-byte form also serialised 128 bias loads per lane).
    const float unscale = a.w_alpha / cv_act_scale(*a.amax);
    const int prows = min(CV_TM, HW - tile * CV_TM);       // valid pixels of this tile
    if (GROUPED) {
        // wave column wn IS the group (64 channels each): slot = (n / cams) * 4 + wn, view = n % cams
        const int bt = n / a.cams, cam = n - bt * a.cams;
        float *obase = a.out + ((((size_t)bt * 4 + wn) * a.cams + cam) * HW + (size_t)tile * CV_TM) * 64;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            const int c = 16 * nn + 4 * lk;
            const cv_f4 bv = a.bias ? *reinterpret_cast<const cv_f4 *>(a.bias + 64 * wn + c) : (cv_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int p = 128 * wm + 16 * m + li;
                if (p < prows)
                    *reinterpret_cast<cv_f4 *>(obase + (size_t)p * 64 + c) = cv_fma4(acc[m][nn], unscale, bv);
            }
        }
        return;
    }
    const float *pmap = dead ? a.pixel_bias_dead : a.pixel_bias;
    const float *pbase = pmap ? pmap + (size_t)tile * CV_TM * CV_COUT : nullptr;
    if (MODE == CV_OUT_Q16) {
        // A (pixel, 64-channel block) of the value stream is pixel li of tile m in wave column wn: its 64 values are the 16 this
        // lane holds (4 tiles nn x 4 registers) and those of the lanes li + 16, li + 32, li + 48 -- block maximum = 16 in-lane
        // values and two cross-lane steps; every lane then quantises its own 16 values (four 8-byte stores).
        short *qbase = a.q + ((size_t)n * HW + (size_t)tile * CV_TM) * CV_COUT + 64 * wn + 4 * lk;
        float *sbase = a.qscale + ((size_t)n * HW + (size_t)tile * CV_TM) * 4 + wn;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            cv_f4 pb[4][4];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                const int p = min(128 * wm + 16 * (4 * half + mm) + li, prows - 1);
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    const int col = 64 * wn + 16 * nn + 4 * lk;
                    pb[mm][nn] = pbase ? *reinterpret_cast<const cv_f4 *>(pbase + (size_t)p * CV_COUT + col)
                                       : (a.bias ? *reinterpret_cast<const cv_f4 *>(a.bias + col) : (cv_f4){0.f, 0.f, 0.f, 0.f});
                }
            }
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                const int m = 4 * half + mm, p = 128 * wm + 16 * m + li;
                cv_f4 v[4];
                unsigned bm = 0u;
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    v[nn] = cv_fma4(acc[m][nn], unscale, pb[mm][nn]);
                    bm = max(bm, rac_absbits4(v[nn][0], v[nn][1], v[nn][2], v[nn][3]));
                }
                bm = max(bm, (unsigned)__shfl_xor((int)bm, 16, 64));
                bm = max(bm, (unsigned)__shfl_xor((int)bm, 32, 64));
                float up, dn;
                rac_q16_factors(bm, up, dn);
                if (p < prows) {
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn)
                        *reinterpret_cast<uint2 *>(qbase + (size_t)p * CV_COUT + 16 * nn) = rac_q16x4(v[nn][0], v[nn][1], v[nn][2], v[nn][3], up);
                    if (lk == 0)
                        sbase[(size_t)p * 4] = dn;
                }
            }
        }
        return;
    }
    float *obase = a.out + ((size_t)n * HW + (size_t)tile * CV_TM) * CV_COUT;
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
        const int col = 64 * wn + 16 * nn + 4 * lk;
        const cv_f4 bv = (!pbase && a.bias) ? *reinterpret_cast<const cv_f4 *>(a.bias + col) : (cv_f4){0.f, 0.f, 0.f, 0.f};
        cv_f4 pb[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int p = min(128 * wm + 16 * m + li, prows - 1);      // (rows past a ragged tile re-read its last row; not stored)
            pb[m] = pbase ? *reinterpret_cast<const cv_f4 *>(pbase + (size_t)p * CV_COUT + col) : bv;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int p = 128 * wm + 16 * m + li;
            if (p < prows)
                *reinterpret_cast<cv_f4 *>(obase + (size_t)p * CV_COUT + col) = cv_fma4(acc[m][nn], unscale, pb[m]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv, stride 2, 64 channels
// The downsample convolution of the temporal encoder (3x3, stride 2, pad 1, 256 -> 64; models/racformer_transformer.py:632,
// 646) on the same activation image (its first Cin/32 chunks; the image may carry more channels: chunks_total) and the
// same arithmetic.  Tile = 256 consecutive output pixels of one image x 64 output channels; 8 waves along the pixels, each
// 32 pixels x 64 channels (2 x 4 MFMA tiles); output NCHW f32 (what the ConvGRU's library convolutions read).
struct ConvS2Args {
    const uint4 *xs;
    const uint4 *ws;      // [9][chunks][64][hi 32 | lo 32]
    const float *bias;
    const float *amax;
    float *out;           // [N][out_ctotal][OH][OW], channels 0..63 written
    int N, H, W, chunks, chunks_total, out_ctotal;
    float w_alpha;
};

#ifndef S2_TM
#define S2_TM 128 /* output pixels per workgroup: 256 workgroups of 4 waves for the 8 x 64 x 64 maps (64: 79 us, 128: 77 us, 256: 86 us) */
#endif
#define S2_THREADS (2 * S2_TM)
#define S2_RSTEP (S2_THREADS / 8) /* tile rows staged per pass */
#define S2_NB (64 / S2_RSTEP)     /* passes for the 64 weight rows */
__global__ __launch_bounds__(S2_THREADS, 2) void conv3x3s2_c64_f16x3_kernel(const ConvS2Args a)
{
    extern __shared__ uint4 lds4[];
    constexpr int STAGE = S2_TM * 8 + 512;   // uint4 per LDS stage: A S2_TM rows, B 64 rows
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int H = a.H, W = a.W, Wp = W + 2, OH = H >> 1, OW = W >> 1, chunks = a.chunks;
    const int tiles_per_img = (OH * OW) / S2_TM;
    const int n = blockIdx.x / tiles_per_img, tile = blockIdx.x - n * tiles_per_img;
    const int KS = 9 * chunks;
    const size_t pix_stride = (size_t)a.chunks_total * 8;
    auto row_base = [&](int j) -> size_t {
        const int r = (tid >> 3) + S2_RSTEP * j;
        const int gp = tile * S2_TM + r, oh = gp / OW, ow = gp - oh * OW;
        return (((size_t)n * (H + 2) + 2 * oh) * Wp + 2 * ow) * pix_stride + (tid & 7);
    };
    auto row_slot = [&](int j) -> int {
        const int r = (tid >> 3) + S2_RSTEP * j;
        return r * 8 + ((tid & 7) ^ ((r >> 1) & 7));
    };
    const size_t a_base0 = row_base(0), a_base1 = row_base(1), a_base2 = row_base(2), a_base3 = row_base(3);
    const int st0 = row_slot(0), st1 = row_slot(1), st2 = row_slot(2), st3 = row_slot(3);
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define S2_GLOAD(ks_)                                                                          \
    do {                                                                                       \
        const int chunk_ = CV_TAP_MAJOR ? (ks_) % chunks : (ks_) / 9;                          \
        const int tap_ = CV_TAP_MAJOR ? (ks_) / chunks : (ks_) - chunk_ * 9;  /* chunk-major, as above */ \
        const int dy_ = tap_ / 3, dx_ = tap_ - dy_ * 3;                                        \
        const size_t off_ = ((size_t)dy_ * Wp + dx_) * pix_stride + (size_t)chunk_ * 8;        \
        const size_t wk_ = (size_t)(tap_ * chunks + chunk_) * 512;                             \
        ra0 = a.xs[a_base0 + off_]; ra1 = a.xs[a_base1 + off_];                                \
        ra2 = a.xs[a_base2 + off_]; ra3 = a.xs[a_base3 + off_];                                \
        rb0 = a.ws[wk_ + tid]; rb1 = a.ws[wk_ + S2_THREADS + tid];                             \
        if (S2_NB > 2) {                                                                       \
            rb2 = a.ws[wk_ + 2 * S2_THREADS + tid];                                            \
            rb3 = a.ws[wk_ + 3 * S2_THREADS + tid];                                            \
        }                                                                                      \
    } while (0)
#define S2_LSTORE(buf_)                                                                        \
    do {                                                                                       \
        uint4 *A_ = lds4 + (buf_) * STAGE, *B_ = A_ + S2_TM * 8;                               \
        A_[st0] = ra0; A_[st1] = ra1; A_[st2] = ra2; A_[st3] = ra3;                            \
        B_[st0] = rb0; B_[st1] = rb1;                                                          \
        if (S2_NB > 2) {                                                                       \
            B_[st2] = rb2; B_[st3] = rb3;                                                      \
        }                                                                                      \
    } while (0)

    cv_f4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
            acc[m][nn] = (cv_f4){0.f, 0.f, 0.f, 0.f};
    int bidx_h[4], bidx_l[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
        const int row = 16 * nn + li, f = (row >> 1) & 7;
        bidx_h[nn] = S2_TM * 8 + row * 8 + (lk ^ f);
        bidx_l[nn] = S2_TM * 8 + row * 8 + ((4 + lk) ^ f);
    }
    const int arow0 = 32 * wave + li;

    S2_GLOAD(0);
    S2_LSTORE(0);
    S2_GLOAD(KS > 1 ? 1 : 0);
    __syncthreads();
    for (int ks = 0; ks < KS; ++ks) {
        S2_LSTORE((ks + 1) & 1);
        const int kn = ks + 2 < KS ? ks + 2 : KS - 1;
        S2_GLOAD(kn);
        __builtin_amdgcn_sched_barrier(0);   // (as in conv3x3_f16x3_kernel: the loads stay at the top of the step)
        const cv_h8 *S = reinterpret_cast<const cv_h8 *>(lds4 + (ks & 1) * STAGE);
        cv_h8 bh[4], bl[4];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            bh[nn] = S[bidx_h[nn]];
            bl[nn] = S[bidx_l[nn]];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = arow0 + 16 * m, f = (row >> 1) & 7;
            const cv_h8 ah = S[row * 8 + (lk ^ f)];
            const cv_h8 al = S[row * 8 + ((4 + lk) ^ f)];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nn], acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nn], acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nn], acc[m][nn], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: NCHW store -- for a channel (column li) the lane's 4 accumulator rows are 4 consecutive output pixels
    const float unscale = a.w_alpha / cv_act_scale(*a.amax);
    float *obase = a.out + (size_t)n * a.out_ctotal * OH * OW + (size_t)tile * S2_TM;
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
        const int co = 16 * nn + li;
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int p = 32 * wave + 16 * m + 4 * lk;
            rac_f4 o = {acc[m][nn][0] * unscale + bv, acc[m][nn][1] * unscale + bv, acc[m][nn][2] * unscale + bv,
                        acc[m][nn][3] * unscale + bv};
            *reinterpret_cast<rac_f4 *>(obase + (size_t)co * OH * OW + p) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ C-ABI
extern "C" int rac_absmax_fwd(const float *const *srcs, const int64_t *counts, int num, float floor_value, float *amax_out,
                              void *stream)
{
    RAC_CHECK_ARG(num >= 0 && amax_out && floor_value >= 0.f, "rac_absmax_fwd: bad arguments");
    unsigned floor_bits;
    memcpy(&floor_bits, &floor_value, sizeof(floor_bits));
#if defined(RAC_ABSMAX_MEMSET) && defined(RAC_DIAGNOSTIC_BUILD)
    // diagnostic builds only (tools/graph_memset_edges.py): round 3's reset of the scale word, a graph MEMSET node when captured
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(amax_out), (int)floor_bits, 1, (hipStream_t)stream);
#else
    hipLaunchKernelGGL(absmax_init_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned *>(amax_out), floor_bits);
#endif
    for (int i = 0; i < num; ++i) {
        RAC_CHECK_ARG(counts[i] >= 0 && (counts[i] == 0 || srcs[i]), "rac_absmax_fwd: source %d", i);
        RAC_CHECK_ARG(((uintptr_t)srcs[i] & 15) == 0, "rac_absmax_fwd: source %d is not 16-byte aligned", i);
        if (counts[i] == 0)
            continue;
        long blocks = (counts[i] / 4 + 255) / 256;
        blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, srcs[i], (long)counts[i],
                           reinterpret_cast<unsigned *>(amax_out));
    }
    return rac_launch_status("rac_absmax_fwd");
}

extern "C" int rac_conv_pack_fwd(const float *src, const float *amax, void *xs, int N, int C, int H, int W, int c_total,
                                 int c_offset, void *stream)
{
    RAC_CHECK_ARG(N >= 0 && C > 0 && C % 32 == 0 && c_total % 32 == 0 && c_offset % 32 == 0 && c_offset + C <= c_total,
                  "rac_conv_pack_fwd: channels C=%d c_total=%d c_offset=%d (multiples of 32)", C, c_total, c_offset);
    RAC_CHECK_ARG(H > 0 && W >= 1 && (long)N * H * (C / 32) * ((W + 127) / 128) < (1l << 31), "rac_conv_pack_fwd: H=%d W=%d", H, W);
    if (N == 0)
        return 0;
    RAC_CHECK_ARG(src && amax && xs, "rac_conv_pack_fwd: null pointer");
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)(N * H * (C / 32) * ((W + 127) / 128))), dim3(256), 0, (hipStream_t)stream, src, (const float *)nullptr,
                       amax, reinterpret_cast<_Float16 *>(xs), C, H, W, c_total / 32, c_offset / 32, 1, 1);
    return rac_launch_status("rac_conv_pack_fwd");
}

extern "C" int rac_conv_pack_bias_fwd(const float *src, const float *bias, const float *amax, void *xs, int N, int C, int H, int W,
                                      int c_total, int c_offset, int frames_per_group, int live_per_group, void *stream)
{
    RAC_CHECK_ARG(N >= 0 && C > 0 && C % 32 == 0 && c_total % 32 == 0 && c_offset % 32 == 0 && c_offset + C <= c_total,
                  "rac_conv_pack_bias_fwd: channels C=%d c_total=%d c_offset=%d (multiples of 32)", C, c_total, c_offset);
    RAC_CHECK_ARG(H > 0 && W >= 1 && (long)N * H * (C / 32) * ((W + 127) / 128) < (1l << 31), "rac_conv_pack_bias_fwd: H=%d W=%d", H, W);
    RAC_CHECK_ARG(frames_per_group >= 1 && live_per_group >= 0 && live_per_group <= frames_per_group && N % frames_per_group == 0,
                  "rac_conv_pack_bias_fwd: N=%d frames in groups of %d, %d live", N, frames_per_group, live_per_group);
    if (N == 0)
        return 0;
    RAC_CHECK_ARG((src || live_per_group == 0) && amax && xs, "rac_conv_pack_bias_fwd: null pointer");
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)(N * H * (C / 32) * ((W + 127) / 128))), dim3(256), 0, (hipStream_t)stream, src, bias, amax,
                       reinterpret_cast<_Float16 *>(xs), C, H, W, c_total / 32, c_offset / 32, frames_per_group, live_per_group);
    return rac_launch_status("rac_conv_pack_bias_fwd");
}

static int cv_launch_plain(const void *xs, const void *ws, const float *bias, const float *pixel_bias, const float *amax, float w_alpha,
                           float *out, void *q, float *qscale, int N, int H, int W, int Cin, int Cout, void *stream, const char *what,
                           const float *pixel_bias_dead = nullptr, int Cin_dead = 0, int frames_per_group = 0, int live_per_group = 0)
{
    RAC_CHECK_ARG(Cout == CV_COUT, "%s: built for %d output channels (got %d)", what, CV_COUT, Cout);
    RAC_CHECK_ARG(Cin > 0 && Cin % 32 == 0, "%s: Cin=%d (multiple of 32)", what, Cin);
    RAC_CHECK_ARG(N >= 0 && H > 0 && W > 0, "%s: N=%d H=%d W=%d", what, N, H, W);
    RAC_CHECK_ARG(!pixel_bias || (H * W) % CV_TM == 0, "%s: a pixel_bias map needs H*W=%d to be a multiple of %d", what, H * W, CV_TM);
    if (N == 0)
        return 0;
    RAC_CHECK_ARG(xs && ws && amax && (out || (q && qscale)), "%s: null pointer", what);
    RAC_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(pixel_bias) |
                    reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(pixel_bias_dead)) & 15) == 0,
                  "%s: out / bias / pixel_bias must be 16-byte aligned", what);
    ConvArgs a;
    a.xs = reinterpret_cast<const uint4 *>(xs);
    a.ws = reinterpret_cast<const uint4 *>(ws);
    a.bias = bias; a.pixel_bias = pixel_bias; a.amax = amax; a.out = out;
    a.q = reinterpret_cast<short *>(q); a.qscale = qscale;
    a.N = N; a.H = H; a.W = W; a.chunks = Cin / 32; a.w_alpha = w_alpha; a.cams = 1;
    a.pixel_bias_dead = pixel_bias_dead; a.fpg = pixel_bias_dead ? frames_per_group : 0; a.live = live_per_group; a.chunks_dead = Cin_dead / 32;
    const int lds = 2 * CV_STAGE_U4 * 16;
    const dim3 grid((unsigned)(N * ((H * W + CV_TM - 1) / CV_TM)));
    if (q) {
        if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_CONV3X3_Q16, reinterpret_cast<const void *>(conv3x3_f16x3_kernel<CV_OUT_Q16>), lds))
            return rc_attr;
        hipLaunchKernelGGL(conv3x3_f16x3_kernel<CV_OUT_Q16>, grid, dim3(512), lds, (hipStream_t)stream, a);
    } else {
        if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_CONV3X3, reinterpret_cast<const void *>(conv3x3_f16x3_kernel<CV_OUT_NHWC>), lds))
            return rc_attr;
        hipLaunchKernelGGL(conv3x3_f16x3_kernel<CV_OUT_NHWC>, grid, dim3(512), lds, (hipStream_t)stream, a);
    }
    return rac_launch_status(what);
}

extern "C" int rac_conv3x3_fwd(const void *xs, const void *ws, const float *bias, const float *pixel_bias, const float *amax,
                               float w_alpha, float *out, int N, int H, int W, int Cin, int Cout, void *stream)
{
    return cv_launch_plain(xs, ws, bias, pixel_bias, amax, w_alpha, out, nullptr, nullptr, N, H, W, Cin, Cout, stream, "rac_conv3x3_fwd");
}

extern "C" int rac_conv3x3_q16_fwd(const void *xs, const void *ws, const float *bias, const float *pixel_bias, const float *amax,
                                   float w_alpha, void *q, float *scale, int N, int H, int W, int Cin, int Cout, void *stream)
{
    RAC_CHECK_ARG(N == 0 || (q && scale), "rac_conv3x3_q16_fwd: null pointer");
    return cv_launch_plain(xs, ws, bias, pixel_bias, amax, w_alpha, nullptr, q, scale, N, H, W, Cin, Cout, stream, "rac_conv3x3_q16_fwd");
}

extern "C" int rac_conv3x3_temporal_fwd(const void *xs, const void *ws, const float *pixel_bias_live, const float *pixel_bias_dead,
                                        const float *amax, float w_alpha, float *out, void *q, float *scale, int N, int H, int W, int Cin,
                                        int Cin_dead, int frames_per_group, int live_per_group, void *stream)
{
    const char *what = "rac_conv3x3_temporal_fwd";
    RAC_CHECK_ARG(frames_per_group >= 1 && live_per_group >= 0 && live_per_group <= frames_per_group && N % frames_per_group == 0,
                  "%s: N=%d images in groups of %d, %d live", what, N, frames_per_group, live_per_group);
    RAC_CHECK_ARG(Cin_dead > 0 && Cin_dead % 32 == 0 && Cin_dead <= Cin, "%s: Cin_dead=%d (a multiple of 32, <= Cin=%d)", what, Cin_dead, Cin);
    RAC_CHECK_ARG(N == 0 || (pixel_bias_live && pixel_bias_dead), "%s: both per-pixel maps are needed", what);
    RAC_CHECK_ARG((out != nullptr) != (q != nullptr && scale != nullptr), "%s: exactly one of out / (q, scale)", what);
    return cv_launch_plain(xs, ws, nullptr, pixel_bias_live, amax, w_alpha, out, q, scale, N, H, W, Cin, CV_COUT, stream, what, pixel_bias_dead,
                           Cin_dead, frames_per_group, live_per_group);
}

extern "C" int rac_fpn_conv_fwd(const void *xs, const void *ws, const float *bias, const float *amax, float w_alpha, float *out,
                                int num_images, int H, int W, int Cin, int num_cams, void *stream)
{
    RAC_CHECK_ARG(Cin > 0 && Cin % 32 == 0, "rac_fpn_conv_fwd: Cin=%d (multiple of 32)", Cin);
    RAC_CHECK_ARG(num_images >= 0 && H > 0 && W > 0 && num_cams >= 1 && num_images % num_cams == 0,
                  "rac_fpn_conv_fwd: %d images of %dx%d from %d cameras (the images must be whole (batch, frame) groups)", num_images, H, W,
                  num_cams);
    if (num_images == 0)
        return 0;
    RAC_CHECK_ARG(xs && ws && amax && out, "rac_fpn_conv_fwd: null pointer");
    RAC_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                  "rac_fpn_conv_fwd: out / bias must be 16-byte aligned");
    ConvArgs a;
    a.xs = reinterpret_cast<const uint4 *>(xs);
    a.ws = reinterpret_cast<const uint4 *>(ws);
    a.bias = bias; a.pixel_bias = nullptr; a.amax = amax; a.out = out;
    a.N = num_images; a.H = H; a.W = W; a.chunks = Cin / 32; a.w_alpha = w_alpha; a.cams = num_cams; a.q = nullptr; a.qscale = nullptr;
    a.pixel_bias_dead = nullptr; a.fpg = 0; a.live = 0; a.chunks_dead = 0;
    const int lds = 2 * CV_STAGE_U4 * 16;
    if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_FPN_CONV, reinterpret_cast<const void *>(conv3x3_f16x3_kernel<CV_OUT_GROUPED>), (int)(lds)))
        return rc_attr;
    hipLaunchKernelGGL(conv3x3_f16x3_kernel<CV_OUT_GROUPED>, dim3((unsigned)(num_images * ((H * W + CV_TM - 1) / CV_TM))), dim3(512), lds,
                       (hipStream_t)stream, a);
    return rac_launch_status("rac_fpn_conv_fwd");
}

extern "C" int rac_conv3x3s2_fwd(const void *xs, const void *ws, const float *bias, const float *amax, float w_alpha,
                                 float *out, int out_channels_total, int N, int H, int W, int Cin, int Cin_image, int Cout,
                                 void *stream)
{
    RAC_CHECK_ARG(out_channels_total >= 64, "rac_conv3x3s2_fwd: out_channels_total=%d", out_channels_total);
    RAC_CHECK_ARG(Cout == 64, "rac_conv3x3s2_fwd: built for 64 output channels (got %d)", Cout);
    RAC_CHECK_ARG(Cin > 0 && Cin % 32 == 0 && Cin_image % 32 == 0 && Cin <= Cin_image, "rac_conv3x3s2_fwd: Cin=%d Cin_image=%d", Cin, Cin_image);
    RAC_CHECK_ARG(N >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ((H / 2) * (W / 2)) % S2_TM == 0,
                  "rac_conv3x3s2_fwd: H=%d W=%d (even, (H/2)*(W/2) a multiple of %d)", H, W, S2_TM);
    if (N == 0)
        return 0;
    RAC_CHECK_ARG(xs && ws && amax && out, "rac_conv3x3s2_fwd: null pointer");
    ConvS2Args a;
    a.xs = reinterpret_cast<const uint4 *>(xs);
    a.ws = reinterpret_cast<const uint4 *>(ws);
    a.bias = bias; a.amax = amax; a.out = out;
    a.N = N; a.H = H; a.W = W; a.chunks = Cin / 32; a.chunks_total = Cin_image / 32; a.out_ctotal = out_channels_total; a.w_alpha = w_alpha;
    const int lds = 2 * (S2_TM * 8 + 512) * 16;
    if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_CONV3X3S2, reinterpret_cast<const void *>(conv3x3s2_c64_f16x3_kernel), (int)(lds)))
        return rc_attr;
    hipLaunchKernelGGL(conv3x3s2_c64_f16x3_kernel, dim3((unsigned)(N * ((H / 2) * (W / 2) / S2_TM))), dim3(S2_THREADS), lds, (hipStream_t)stream, a);
    return rac_launch_status("rac_conv3x3s2_fwd");
}
