// conv_direct.hip -- the ConvGRU branch of RadarBEVTemporalEncoder (models/racformer_transformer.py:645-656, 674-720) on own
// kernels (gfx950): the stride-2 downsample 256 -> 64, the gates convolution's x half 64 -> 192 for the four live frames in one
// launch, the recurrence's h half 64 -> 192 with the whole GRU update in its epilogue (one launch per step), and the 64 -> 64
// convolution behind the 2x resize, written straight into the temporal-fusion convolution's activation image.
//
// Rounds 2-4 ran these as one LDS-tiled kernel (conv3x3s2_c64_f16x3_kernel, 55 us) plus five MIOpen Winograd launches, four gate
// kernels, a resize and a pack (211 us): 64 x 64-pixel maps, strictly sequential, ~1 us per K step of global -> LDS -> barrier ->
// a few MFMAs.  What these launches need is a short dependent chain, not throughput.  The shape here:
//   * WEIGHTS through LDS, a whole 32-channel chunk (9 taps) of the workgroup's output-channel block at a time (LDS-DMA,
//     global_load_lds_dwordx4: no staging registers), two chunk buffers -- with 64 input channels (the GRU, the resize
//     convolution) everything a workgroup will ever multiply by is resident before its first MFMA, and the K loop runs with ONE
//     workgroup barrier in front of it; the 256-channel downsample pays one barrier per chunk (8), not per K step (72);
//   * PIXELS never pass through LDS: the activation images hold per pixel and 32-channel chunk a 128-byte line [hi 32 | lo 32]
//     f16, and a lane's MFMA B fragment is 16 bytes of it -- every lane loads its own fragments, D K-steps ahead through a
//     register ring with compile-time slots (the K loop is straight-line code: counted vmcnt waits, never a drain);
//     A lane's fragment belongs to pixel lane & 15, K group lane >> 4, so in that order FOUR CONSECUTIVE lanes read four different
//     pixels' lines and the texture addresser, which coalesces a quad of lanes, sees 64 separate 16-byte requests per instruction
//     (measured: 10 B / clock / CU, a third of the L2 gather rate).  The loads are therefore issued in line order -- lane = 4 pixel
//     + K group: a quad reads 64 contiguous bytes -- and brought into fragment order by ds_bpermute (8 per pixel tile and K step,
//     a step ahead of their MFMAs: the crossbar, no LDS memory);
//   * the first version of this file (round 5) loaded the weight fragments per wave from global memory as well: every wave then
//     pulls the whole weight block through the CU's texture path (6 KB per K step against 2 KB of pixels) -- 21 us per GRU step,
//     96 us for the downsample, slower than what it replaced.  The texture path is the scarce resource of a CU.
//
// Arithmetic as conv3x3.hip: v * 2^e = hi + lo (two f16), products hi*lo + lo*hi + hi*hi on v_mfma_f32_16x16x32_f16, fp32
// accumulate (truncation 2^-22).  The activations' 2^e never needs a pass over the data: |ConvGRU state| <= 1 (convex combinations
// of tanh values, zero start: bilinear resizing keeps that), and |downsample(x)| <= max_row ||W||_1 max|x| + max|b| follows from the
// input maximum rac_absmax_fwd already measured (a loose bound costs nothing: hi / lo are floating point, values down to 2^-13 of
// the bound keep their 22 bits).
// The weights are the MFMA's A operand, the pixels its B operand: an accumulator tile has the PIXEL on the lane (column li) and
// four consecutive output CHANNELS in the lane's registers (rows 4 lk + r) -- channel-last 16-byte stores, and for the GRU the
// z, r and candidate pre-activations of one (pixel, channel) in the same lane (tiles 0 / 1 / 2 = the three gate blocks).
#include "rac_common.h"

typedef _Float16 cd_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cd_h4 __attribute__((ext_vector_type(4)));
typedef float cd_f4 __attribute__((ext_vector_type(4)));

struct CdArgs {
    const uint4 *xs;
    const uint4 *ws;
    int N, Hi, Wi, OH, OW;
    int in_chunks_total, in_chunk0, Cout;
    rac_cd_frames in_frames, out_frames, xpart_frames, h_prev_frames, h_out_frames;
    rac_cd_scale in_scale, out_scale;
    float w_alpha;
    const float *bias;
    _Float16 *out_img;
    int out_chunks_total, out_chunk0;
    float *out_f32;
    const float *pixel_map;
    const float *xpart;
    const float *h_prev;
    float *h_out;
};

__device__ __forceinline__ int cd_frame(const rac_cd_frames &f, int n)
{
    return (n / f.live) * f.stride + n % f.live + f.first;
}
__device__ __forceinline__ float cd_scale(const rac_cd_scale &s)
{
    return rac_act_scale((s.amax ? s.mul * *s.amax : 0.f) + s.add);
}

// Workgroup = 4 waves = 64 PT consecutive output pixels of one frame x one block of 16 NT output channels.
//   NT  16-channel accumulator tiles per wave (A operand tiles, read from LDS)
//   PT  16-pixel tiles per wave (B operand tiles, loaded from the image): every A fragment read from LDS feeds PT MFMAs
//   D   K steps whose pixel fragments are in flight (divides 9)
//   KCH 32-channel chunks of the input = 9 KCH K steps, a compile-time constant: the K loop is straight-line code, which is what
//       keeps the ring in fixed registers (as a run-time loop hipcc rotates the ring's registers across the back edge with copies,
//       and a copy of a register that a load is still writing is a vmcnt(0) at the top of every trip).  KCH = 0: no convolution.
// LDS: two chunk buffers [9 taps][NT tiles][16 rows][8 slots of 16 B], slot s of row r at position s ^ ((r >> 1) & 7) (16-byte
// fragment reads without bank conflicts; LDS-DMA writes a wave-instruction's 1 KB linearly, so the permutation sits on the source
// side, as in gemm_split.hip).
template <int STRIDE, int NT, int PT, int MODE, int D, int KCH>
__global__ __launch_bounds__(256) void conv_direct_kernel(const CdArgs a)
{
    static_assert(9 % D == 0, "ring depth must divide the 9 taps of a chunk");
    static_assert(MODE != RAC_CD_GRU || NT == 3, "GRU mode: tiles = (z, r, candidate)");
    extern __shared__ uint4 cd_lds[];
    constexpr int SLICE = 9 * NT * 128;                    // uint4 per chunk buffer
    constexpr int KS = 9 * KCH;
    constexpr int PIECES = 18 * NT;                        // 1 KB LDS-DMA pieces per chunk: (tap, tile, half of its 16 rows)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lk = lane >> 4;
    const int npix = a.OH * a.OW, ptiles = (npix + 64 * PT - 1) / (64 * PT);
    const int cblocks = MODE == RAC_CD_GRU ? 4 : a.Cout / (16 * NT);
    int bid = blockIdx.x;
    const int cb = bid % cblocks;
    bid /= cblocks;
    const int pt = bid % ptiles, n = bid / ptiles;
    // first output channel of accumulator tile nn
    int co0[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn)
        co0[nn] = MODE == RAC_CD_GRU ? 64 * nn + 16 * cb : (cb * NT + nn) * 16;
    // this lane's pixels: as accumulator column li of pixel tile j (epilogue), and as LOADER of pixel lane >> 2, K group lane & 3
    // (line order: the four lanes of a quad read 64 contiguous bytes of one pixel's line)
    int pix[PT];
    const uint4 *xb[PT];
    const int Wp = a.Wi + 2;
    const size_t pix_u4 = (size_t)a.in_chunks_total * 8;
    const size_t fin = (size_t)(KCH > 0 ? cd_frame(a.in_frames, n) : 0) * (a.Hi + 2);
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int p0 = pt * (64 * PT) + wave * (16 * PT) + 16 * j;
        pix[j] = p0 + li;
        const int pc = min(p0 + (lane >> 2), npix - 1), oh = pc / a.OW, ow = pc - oh * a.OW;
        xb[j] = a.xs + ((fin + STRIDE * oh) * Wp + STRIDE * ow) * pix_u4 + (size_t)a.in_chunk0 * 8 + (lane & 3);
    }
    // fragment order <- line order: lane (li, lk) takes what loader lane 4 li + lk fetched
    const int perm_addr = ((lane & 15) * 4 + (lane >> 4)) * 4;

    cd_f4 acc[NT][PT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn)
#pragma unroll
        for (int j = 0; j < PT; ++j)
            acc[nn][j] = (cd_f4){0.f, 0.f, 0.f, 0.f};

    if constexpr (KCH > 0) {
        // weight chunk c -> buffer c & 1: piece q = (tap, tile, row half); lane = (row lane >> 3 of the half, LDS slot lane & 7)
        const int rrow = lane >> 3, rslot = lane & 7;
        // LDS-DMA as buffer_load ... lds (a buffer descriptor over the weight image), NOT global_load_lds: hipcc books the global form as a
        // FLAT access that may return out of order, and every vector-memory wait behind one becomes vmcnt(0) -- the 256-channel
        // downsample, which issues chunk c + 2's pieces inside its K loop, drained its pixel ring and the pieces at every chunk
        // boundary (ISA reading; the buffer form gets counted waits).  The piece loop is unrolled: a loop of unknown trip count
        // between a load and its use costs the same vmcnt(0).
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(a.ws), 0, 9 * KCH * a.Cout * 128, 0x00020000);
        auto dma_chunk = [&](int c) {
            char *dst = reinterpret_cast<char *>(cd_lds + (c & 1) * SLICE);
#pragma unroll
            for (int i = 0; i < (PIECES + 3) / 4; ++i) {
                const int q = wave + 4 * i;
                if (PIECES % 4 == 0 || q < PIECES) {
                    const int tap = q / (2 * NT), rem = q - tap * (2 * NT), nn = rem >> 1, half = rem & 1;
                    const int row = 8 * half + rrow;
                    const int co = MODE == RAC_CD_GRU ? 64 * nn + 16 * cb + row : (cb * NT + nn) * 16 + row;
                    const unsigned voff = (unsigned)((((tap * KCH + c) * a.Cout + co) * 8 + (rslot ^ ((row >> 1) & 7))) * 16);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(dst + ((tap * NT + nn) * 128 + half * 64) * 16),
                                                             16, voff, 0, 0, 0);
                }
            }
        };
        // A fragments of K step (chunk c, tap): row li of tile nn, hi slot lk, lo slot 4 + lk
        const int fsw = (li >> 1) & 7, ahi = li * 8 + (lk ^ fsw), alo = li * 8 + ((4 + lk) ^ fsw);
        uint4 xr[D][PT][2];
        cd_h8 wh[2][NT], wl[2][NT];
#define CD_XLOAD(slot_, ks_)                                                                       \
    do {                                                                                           \
        constexpr int ksc_ = (ks_) < KS ? (ks_) : KS - 1;     /* past the end: re-fetch the last step */ \
        constexpr int chunk_ = ksc_ / 9, tap_ = ksc_ % 9, dy_ = tap_ / 3, dx_ = tap_ % 3;         \
        const size_t off_ = ((size_t)dy_ * Wp + dx_) * pix_u4 + (size_t)chunk_ * 8;                \
        _Pragma("unroll") for (int j = 0; j < PT; ++j)                                             \
        {                                                                                          \
            xr[slot_][j][0] = xb[j][off_];                                                         \
            xr[slot_][j][1] = xb[j][off_ + 4];                                                     \
        }                                                                                          \
    } while (0)
#define CD_ALOAD(set_, ks_)                                                                        \
    do {                                                                                           \
        constexpr int kk_ = (ks_) < KS ? (ks_) : KS - 1;                                           \
        const cd_h8 *S_ = reinterpret_cast<const cd_h8 *>(cd_lds + ((kk_ / 9) & 1) * SLICE + (kk_ % 9) * NT * 128); \
        _Pragma("unroll") for (int nn = 0; nn < NT; ++nn)                                          \
        {                                                                                          \
            wh[set_][nn] = S_[nn * 128 + ahi];                                                     \
            wl[set_][nn] = S_[nn * 128 + alo];                                                     \
        }                                                                                          \
    } while (0)

        uint4 xq[2][PT][2];                                 // pixel fragments in fragment order, this step's and the next one's
#define CD_XPERM(set_, slot_)                                                                      \
    do {                                                                                           \
        _Pragma("unroll") for (int j = 0; j < PT; ++j) _Pragma("unroll") for (int hl = 0; hl < 2; ++hl) \
        {                                                                                          \
            xq[set_][j][hl].x = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)xr[slot_][j][hl].x); \
            xq[set_][j][hl].y = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)xr[slot_][j][hl].y); \
            xq[set_][j][hl].z = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)xr[slot_][j][hl].z); \
            xq[set_][j][hl].w = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)xr[slot_][j][hl].w); \
        }                                                                                          \
    } while (0)

        rac_static_for<0, D>([&](auto d) { CD_XLOAD(d.value, d.value); });     // (ahead of the weight pieces: they overlap)
        dma_chunk(0);
        if (KCH > 1)
            dma_chunk(1);
        __builtin_amdgcn_s_waitcnt(0x0F70 | 0);            // vmcnt(0): once, in front of the loop (weights and the ring's first loads)
        __syncthreads();
        CD_ALOAD(0, 0);
        CD_XPERM(0, 0);
        CD_XLOAD(0, D);
        __builtin_amdgcn_sched_barrier(0);
        rac_static_for<0, KS>([&](auto ks_c) {
            constexpr int ks = ks_c.value, set = ks & 1;
            if constexpr (ks % 9 == 8 && ks + 1 < KS) {
                // chunk boundary, taken one step early (the A fragments of the next chunk's first step are read below): every wave
                // is done reading this chunk's buffer (its last fragments were read a step ago), and the next chunk's pieces --
                // issued a chunk ago, older than everything in the ring -- have landed
                constexpr int c = ks / 9;
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D * PT * 2) : "memory");
                __syncthreads();
                if constexpr (c + 2 < KCH)
                    dma_chunk(c + 2);                       // into the buffer chunk c leaves
            }
            if constexpr (ks + 1 < KS) {
                CD_ALOAD(set ^ 1, ks + 1);                  // next step's weight fragments and
                CD_XPERM(set ^ 1, (ks + 1) % D);            // pixel fragments under this step's MFMAs;
                CD_XLOAD((ks + 1) % D, ks + 1 + D);         // the ring slot they leave takes K step ks + 1 + D
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const cd_h8 bh = __builtin_bit_cast(cd_h8, xq[set][j][0]), bl = __builtin_bit_cast(cd_h8, xq[set][j][1]);
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    acc[nn][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[set][nn], bl, acc[nn][j], 0, 0, 0);
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    acc[nn][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[set][nn], bh, acc[nn][j], 0, 0, 0);
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    acc[nn][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[set][nn], bh, acc[nn][j], 0, 0, 0);
            }
            // (nothing moves across: with the issue order pinned the compiler's waits are counted -- the pixel fragments of the
            //  D - 1 younger steps stay in flight)
            __builtin_amdgcn_sched_barrier(0);
        });
#undef CD_XPERM
#undef CD_XLOAD
#undef CD_ALOAD
    }

    const float unscale = KCH > 0 ? a.w_alpha / cd_scale(a.in_scale) : 0.f;
    const cd_f4 us4 = {unscale, unscale, unscale, unscale};
    const float so = MODE == RAC_CD_F32 ? 1.f : cd_scale(a.out_scale);
    const size_t fout = MODE == RAC_CD_F32 ? 0 : (size_t)cd_frame(a.out_frames, n) * (a.OH + 2);
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int p = pix[j];
        if (p >= npix)
            continue;
        if (MODE == RAC_CD_F32) {
            float *o = a.out_f32 + ((size_t)n * npix + p) * a.Cout;
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
                const int c = co0[nn] + 4 * lk;
                cd_f4 add = a.bias ? *reinterpret_cast<const cd_f4 *>(a.bias + c) : (cd_f4){0.f, 0.f, 0.f, 0.f};
                if (a.pixel_map)
                    add += *reinterpret_cast<const cd_f4 *>(a.pixel_map + (size_t)p * a.Cout + c);
                *reinterpret_cast<cd_f4 *>(o + c) = __builtin_elementwise_fma(acc[nn][j], us4, add);
            }
            continue;
        }
        // the activation-image destination: pixel (oh + 1, ow + 1) of the output frame, per 32-channel chunk [hi 32 | lo 32]
        const int oh = p / a.OW, ow = p - oh * a.OW;
        _Float16 *opix = a.out_img + (((fout + oh + 1) * (a.OW + 2) + ow + 1) * a.out_chunks_total + a.out_chunk0) * 64;
        auto store_img = [&](int co, cd_f4 v) {      // channels co .. co + 3 of this launch's output
            cd_h4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = v[r] * so;
                hi[r] = (_Float16)y;
                lo[r] = (_Float16)(y - (float)hi[r]);
            }
            _Float16 *d = opix + (co >> 5) * 64 + (co & 31);
            *reinterpret_cast<cd_h4 *>(d) = hi;
            *reinterpret_cast<cd_h4 *>(d + 32) = lo;
        };
        if (MODE == RAC_CD_IMAGE) {
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
                const int c = co0[nn] + 4 * lk;
                const cd_f4 bv = a.bias ? *reinterpret_cast<const cd_f4 *>(a.bias + c) : (cd_f4){0.f, 0.f, 0.f, 0.f};
                store_img(c, __builtin_elementwise_fma(acc[nn][j], us4, bv));
            }
            continue;
        }
        // GRU update (models/racformer_transformer.py:714-720): channel c = 16 cb + 4 lk + r of the 64
        const int c4 = 16 * cb + 4 * lk;
        const float *xp = a.xpart + ((size_t)cd_frame(a.xpart_frames, n) * npix + p) * 192 + c4;
        const cd_f4 xz = *reinterpret_cast<const cd_f4 *>(xp), xg = *reinterpret_cast<const cd_f4 *>(xp + 64),
                    xc = *reinterpret_cast<const cd_f4 *>(xp + 128);
        const cd_f4 hp = a.h_prev ? *reinterpret_cast<const cd_f4 *>(a.h_prev + ((size_t)cd_frame(a.h_prev_frames, n) * npix + p) * 64 + c4)
                                  : (cd_f4){0.f, 0.f, 0.f, 0.f};
        cd_f4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float zg = __builtin_fmaf(acc[0][j][r], unscale, xz[r]);
            const float rg = __builtin_fmaf(acc[1][j][r], unscale, xg[r]);
            const float cg = __builtin_fmaf(acc[2][j][r], unscale, xc[r]);
            const float z = 1.f / (1.f + expf(-zg));
            const float rr = 1.f / (1.f + expf(-rg));
            const float cand = tanhf(cg + rr * hp[r]);
            h[r] = (1.f - z) * hp[r] + z * cand;
        }
        *reinterpret_cast<cd_f4 *>(a.h_out + ((size_t)cd_frame(a.h_out_frames, n) * npix + p) * 64 + c4) = h;
        store_img(c4, h);
    }
}

// nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) of channel-last maps into an activation image
// (models/racformer_transformer.py:633-636; torch's source index with align_corners: src = dst * (in - 1) / (out - 1), the same
// blend order as upsample2x_kernel of temporal.hip).  One thread per (output pixel, 8 channels): a 16-byte hi and a 16-byte lo store.
__global__ __launch_bounds__(256) void upsample2x_image_kernel(const float *__restrict__ src, _Float16 *__restrict__ img, long frames,
                                                               int h, int w, int C, float so)
{
    const int oh = 2 * h, ow = 2 * w, c8n = C >> 3, chunks = C >> 5;
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f;
    const float sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    const long n = frames * oh * ow * c8n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        long t = i / c8n;
        const int x = (int)(t % ow);
        t /= ow;
        const int y = (int)(t % oh);
        const long f = t / oh;
        const float fy = sy * (float)y, fx = sx * (float)x;
        const int y0 = (int)fy, y1 = min(y0 + 1, h - 1), x0 = (int)fx, x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, hy = 1.f - ly, lx = fx - (float)x0, hx = 1.f - lx;
        const float *b = src + (size_t)f * h * w * C + c8 * 8;
        const float *p00 = b + ((size_t)y0 * w + x0) * C, *p01 = b + ((size_t)y0 * w + x1) * C;
        const float *p10 = b + ((size_t)y1 * w + x0) * C, *p11 = b + ((size_t)y1 * w + x1) * C;
        cd_h8 hi, lo;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const rac_f4 a00 = rac_ld4(p00 + 4 * q), a01 = rac_ld4(p01 + 4 * q), a10 = rac_ld4(p10 + 4 * q), a11 = rac_ld4(p11 + 4 * q);
            const float v[4] = {hy * (hx * a00.x + lx * a01.x) + ly * (hx * a10.x + lx * a11.x),
                                hy * (hx * a00.y + lx * a01.y) + ly * (hx * a10.y + lx * a11.y),
                                hy * (hx * a00.z + lx * a01.z) + ly * (hx * a10.z + lx * a11.z),
                                hy * (hx * a00.w + lx * a01.w) + ly * (hx * a10.w + lx * a11.w)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = v[r] * so;
                hi[4 * q + r] = (_Float16)s;
                lo[4 * q + r] = (_Float16)(s - (float)hi[4 * q + r]);
            }
        }
        _Float16 *d = img + ((((size_t)f * (oh + 2) + y + 1) * (ow + 2) + x + 1) * chunks + (c8 >> 2)) * 64 + (c8 & 3) * 8;
        *reinterpret_cast<cd_h8 *>(d) = hi;
        *reinterpret_cast<cd_h8 *>(d + 32) = lo;
    }
}

static bool cd_host_scale_ok(const rac_cd_scale &s) { return s.mul >= 0.f && s.add >= 0.f; }

template <int STRIDE, int NT, int PT, int MODE, int KCH>
static int cd_launch(const CdArgs &a, int attr_id, hipStream_t st)
{
    // ring depth 3 (measured at 1 / 3 / 9 with tools/exp_convdirect.py: 39.9 / 31.7 / 31.1 us for the downsample, no difference elsewhere)
    constexpr int D = 3;
    constexpr int lds = KCH > 0 ? 2 * 9 * NT * 128 * 16 : 0;
    const void *fn = reinterpret_cast<const void *>(conv_direct_kernel<STRIDE, NT, PT, MODE, D, KCH>);
    if (lds > 48 * 1024)
        if (const int rc = rac_set_dynamic_lds_once(attr_id, fn, lds))
            return rc;
    const int npix = a.OH * a.OW, ptiles = (npix + 64 * PT - 1) / (64 * PT);
    const int cblocks = MODE == RAC_CD_GRU ? 4 : a.Cout / (16 * NT);
    hipLaunchKernelGGL((conv_direct_kernel<STRIDE, NT, PT, MODE, D, KCH>), dim3((unsigned)(a.N * ptiles * cblocks)), dim3(256), lds, st, a);
    return 0;
}

extern "C" int rac_conv_direct_fwd(const rac_conv_direct *d, void *stream)
{
    RAC_CHECK_ARG(d, "rac_conv_direct_fwd: null descriptor");
    RAC_CHECK_ARG(d->mode >= RAC_CD_IMAGE && d->mode <= RAC_CD_GRU && (d->conv_stride == 1 || d->conv_stride == 2),
                  "rac_conv_direct_fwd: mode=%d conv_stride=%d", d->mode, d->conv_stride);
    RAC_CHECK_ARG(d->N >= 0 && d->H > 0 && d->W > 0 && d->H % d->conv_stride == 0 && d->W % d->conv_stride == 0,
                  "rac_conv_direct_fwd: N=%d H=%d W=%d", d->N, d->H, d->W);
    RAC_CHECK_ARG(d->chunks >= 0 && d->in_chunk0 >= 0 && d->in_chunk0 + d->chunks <= d->in_chunks_total,
                  "rac_conv_direct_fwd: chunks %d..+%d of %d", d->in_chunk0, d->chunks, d->in_chunks_total);
    RAC_CHECK_ARG(d->in_frames.live >= 1 && d->out_frames.live >= 1 && d->xpart_frames.live >= 1 && d->h_prev_frames.live >= 1 &&
                      d->h_out_frames.live >= 1, "rac_conv_direct_fwd: frame maps need live >= 1");
    RAC_CHECK_ARG(cd_host_scale_ok(d->in_scale) && cd_host_scale_ok(d->out_scale), "rac_conv_direct_fwd: scale constants must be >= 0");
    if (d->N == 0)
        return 0;
    RAC_CHECK_ARG(d->chunks == 0 || (d->in_img && d->ws), "rac_conv_direct_fwd: null image / weights");
    RAC_CHECK_ARG(d->chunks > 0 || d->mode == RAC_CD_GRU, "rac_conv_direct_fwd: chunks == 0 is the GRU's first step only");
    CdArgs a;
    a.xs = reinterpret_cast<const uint4 *>(d->in_img);
    a.ws = reinterpret_cast<const uint4 *>(d->ws);
    a.N = d->N; a.Hi = d->H; a.Wi = d->W; a.OH = d->H / d->conv_stride; a.OW = d->W / d->conv_stride;
    a.in_chunks_total = d->in_chunks_total; a.in_chunk0 = d->in_chunk0; a.Cout = d->Cout;
    a.in_frames = d->in_frames; a.out_frames = d->out_frames; a.xpart_frames = d->xpart_frames;
    a.h_prev_frames = d->h_prev_frames; a.h_out_frames = d->h_out_frames;
    a.in_scale = d->in_scale; a.out_scale = d->out_scale; a.w_alpha = d->w_alpha; a.bias = d->bias;
    a.out_img = reinterpret_cast<_Float16 *>(d->out_img); a.out_chunks_total = d->out_chunks_total; a.out_chunk0 = d->out_chunk0;
    a.out_f32 = d->out_f32; a.pixel_map = d->pixel_map; a.xpart = d->xpart; a.h_prev = d->h_prev; a.h_out = d->h_out;
    hipStream_t st = (hipStream_t)stream;
    int rc = -1;
    const int k = d->chunks;
    // The chunk count is a template parameter (straight-line K loop); instantiated: what the encoder uses (64 hidden channels = 2 chunks
    // everywhere, the 256-channel downsample = 8) and a few more for other widths.  Pixels per wave (PT): 16 for the GRU step and the
    // downsample (launches of 4096 / 16384 pixels: 256 workgroups), 32 for the gates' x half, 64 for the resize convolution (65536 pixels).
    if (d->mode == RAC_CD_GRU) {
        RAC_CHECK_ARG(d->Cout == 192 && d->conv_stride == 1, "rac_conv_direct_fwd: the GRU mode is built for 3 x 64 gate channels, stride 1");
        RAC_CHECK_ARG(d->xpart && d->h_out && d->out_img && d->out_chunk0 >= 0 && d->out_chunk0 + 2 <= d->out_chunks_total,
                      "rac_conv_direct_fwd: GRU needs xpart, h_out and an output image with room for 64 channels");
        RAC_CHECK_ARG(((reinterpret_cast<uintptr_t>(d->xpart) | reinterpret_cast<uintptr_t>(d->h_prev) | reinterpret_cast<uintptr_t>(d->h_out)) & 15) == 0,
                      "rac_conv_direct_fwd: xpart / h_prev / h_out must be 16-byte aligned");
        RAC_CHECK_ARG(k == 0 || k == 2, "rac_conv_direct_fwd: the GRU mode is built for 64 hidden channels (chunks = 2; 0 for the first step), got %d", k);
        rc = k == 0 ? cd_launch<1, 3, 1, RAC_CD_GRU, 0>(a, 0, st) : cd_launch<1, 3, 1, RAC_CD_GRU, 2>(a, RAC_ATTR_CD_GRU, st);
        return rc ? rc : rac_launch_status("rac_conv_direct_fwd(gru)");
    }
    RAC_CHECK_ARG(d->Cout > 0 && d->Cout % 64 == 0, "rac_conv_direct_fwd: Cout=%d (a multiple of 64)", d->Cout);
    RAC_CHECK_ARG(((reinterpret_cast<uintptr_t>(d->bias) | reinterpret_cast<uintptr_t>(d->pixel_map) | reinterpret_cast<uintptr_t>(d->out_f32)) & 15) == 0,
                  "rac_conv_direct_fwd: bias / pixel_map / out_f32 must be 16-byte aligned");
    if (d->mode == RAC_CD_IMAGE) {
        RAC_CHECK_ARG(d->out_img && d->out_chunk0 >= 0 && d->out_chunk0 * 32 + d->Cout <= d->out_chunks_total * 32,
                      "rac_conv_direct_fwd: output image has no room for %d channels from chunk %d", d->Cout, d->out_chunk0);
        if (d->conv_stride == 2) {
            RAC_CHECK_ARG(k == 8 || k == 2 || k == 3, "rac_conv_direct_fwd: stride 2 is instantiated for 2 / 3 / 8 input chunks, got %d", k);
            rc = k == 8 ? cd_launch<2, 4, 1, RAC_CD_IMAGE, 8>(a, RAC_ATTR_CD_S2_8, st)
                        : (k == 2 ? cd_launch<2, 4, 1, RAC_CD_IMAGE, 2>(a, RAC_ATTR_CD_S2_2, st) : cd_launch<2, 4, 1, RAC_CD_IMAGE, 3>(a, RAC_ATTR_CD_S2_3, st));
        } else {
            RAC_CHECK_ARG(k == 2, "rac_conv_direct_fwd: stride-1 image mode is instantiated for 2 input chunks, got %d", k);
            rc = cd_launch<1, 4, 4, RAC_CD_IMAGE, 2>(a, RAC_ATTR_CD_IMG_2, st);
        }
        return rc ? rc : rac_launch_status("rac_conv_direct_fwd(image)");
    }
    RAC_CHECK_ARG(d->out_f32 && d->conv_stride == 1, "rac_conv_direct_fwd: the f32 mode needs out_f32 and stride 1");
    RAC_CHECK_ARG(k == 1 || k == 2 || k == 4, "rac_conv_direct_fwd: the f32 mode is instantiated for 1 / 2 / 4 input chunks, got %d", k);
    rc = k == 2 ? cd_launch<1, 4, 2, RAC_CD_F32, 2>(a, RAC_ATTR_CD_F32_2, st)
                : (k == 1 ? cd_launch<1, 4, 2, RAC_CD_F32, 1>(a, RAC_ATTR_CD_F32_1, st) : cd_launch<1, 4, 2, RAC_CD_F32, 4>(a, RAC_ATTR_CD_F32_4, st));
    return rc ? rc : rac_launch_status("rac_conv_direct_fwd(f32)");
}

extern "C" int rac_upsample2x_image_fwd(const float *src, void *img, int frames, int h, int w, int C, float bound, void *stream)
{
    RAC_CHECK_ARG(frames >= 0 && h > 0 && w > 0 && C > 0 && C % 32 == 0 && bound >= 0.f, "rac_upsample2x_image_fwd: frames=%d h=%d w=%d C=%d", frames,
                  h, w, C);
    if (frames == 0)
        return 0;
    RAC_CHECK_ARG(src && img && (reinterpret_cast<uintptr_t>(src) & 15) == 0, "rac_upsample2x_image_fwd: null / unaligned pointer");
    // (host copy of rac_act_scale)
    float so = 1.f;
    if (bound > 1.0e-30f && bound < 3.0e38f) {
        int ex;
        frexpf(bound, &ex);
        so = ldexpf(1.f, 14 - ex);
    }
    long blocks = ((long)frames * 4 * h * w * (C / 8) + 255) / 256;
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(upsample2x_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<_Float16 *>(img),
                       (long)frames, h, w, C, so);
    return rac_launch_status("rac_upsample2x_image_fwd");
}
