// gemm_split.hip -- split-precision GEMMs of AdaptiveMixing on the f16 matrix cores (gfx950), hand-written:
//   rac_outproj_fwd  : out_proj, nn.Linear(32768 -> 256) over 900 queries (models/racformer_transformer.py:566,606), as a
//                      split-K GEMM  partial[s] = Z[:, slice s] @ W[:, slice s]^T
//   rac_generator_fwd: parameter_generator, nn.Linear(256 -> 65536) (models/racformer_transformer.py:565,589)
//
// Arithmetic (same as conv3x3.hip): every operand is v * 2^e = hi + lo (two f16, 22 significant bits); the three
// leading products lo*hi + hi*lo + hi*hi are accumulated in fp32 by v_mfma_f32_16x16x32_f16 -- the dropped lo*lo term
// is 2^-22 relative, the result matches an fp32 GEMM to fp32 rounding.
//
// Operand images (one format for both operands): row r, per 32 values of K one 128-byte line [hi 32 | lo 32] f16.
//   X image  [M][K/32][hi 32 | lo 32]  activations: written by the producing kernel (rac_mixing_fwd's out_split: 118 MB for
//                                       900 x 32768; rac_rowgemm_fwd's split_out: 0.9 MB for 900 x 256)
//   W image  [N][K/32][hi 32 | lo 32]  packed once from the nn.Linear weight [N][K]
// Compared with the K-concatenated [hi | hi | lo] x [hi | lo | hi] operands a library GEMM needs, each value is stored
// once (4 B instead of 6 B).
//
// Workgroup = 768 threads = 8 multiplying waves + 4 loader waves; tile = 256 output features (W rows) x up to 128 rows of
// X, K in steps of 32.  The multiplying waves (4 along the features x 2 along the rows, 64 x 64 = 4 x 4 MFMA tiles each,
// two per SIMD) never issue a vector-memory instruction; the loader waves move each step's 256 + 128 lines (48 KB) global
// -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of three stages (144 KB), two
// steps ahead of the MFMAs (counted vmcnt, one barrier per step).  Why this shape: an LDS-DMA piece costs its wave 60-185
// issue cycles and a CU takes in at most ~68 GB/s this way (MI355X_MICROARCH.md); a 128 x 128 tile with the loads in the
// multiplying waves ran out_proj at 2.2 us per 64-deep step (70 us per launch, as hipBLASLt), with loader waves at 50 us;
// 48 KB for 96 MFMAs per SIMD balances the intake (0.7 us) with the matrix pipe (0.73 us).
// LDS-DMA writes a wave-instruction's 1 KB linearly (8 lines), so bank conflicts are avoided on the SOURCE side: LDS slot
// p of line r receives the line's 16-byte chunk p ^ ((r >> 1) & 7), and the fragment reads apply the same XOR -- every
// ds_read_b128 is conflict-free.
// The MFMA's A operand is W (features), its B operand X (rows): a lane's four accumulator registers are four consecutive
// FEATURES of one row, i.e. one 16-byte store.
// The rows of X are cut into ceil(M/128) tiles of whole 16-row MFMA tiles, as equal as possible (900 rows: seven tiles of
// 112 rows and one of 116), and MFMA tiles past a row tile's end are skipped: no work on padding.
// A workgroup walks `tiles_per_wg` consecutive tiles (row tile fastest) with the ring running across tile boundaries: the
// loaders fetch the next tile while the multiplying waves store the finished one.  Workgroups b, b + 8, ... share an XCD
// (round-robin dispatch) and take consecutive tile ranges, so tiles that share W rows pull them through one L2.
#include "rac_common.h"

typedef _Float16 gs_h8 __attribute__((ext_vector_type(8)));
typedef float gs_f4 __attribute__((ext_vector_type(4)));

#define GS_TW 256                          /* W rows (output features) per tile */
#define GS_TX 128                          /* X rows per tile (at most) */
#define GS_LINE 128                        /* bytes per row and K step of 32: [hi 32 | lo 32] f16 */
#define GS_STAGE ((GS_TW + GS_TX) * GS_LINE) /* 48 KB: W lines then X lines */
#define GS_STAGES 3
#define GS_PIECES_PER_LOADER 12            /* 48 one-KB LDS-DMA pieces per stage, four loader waves */

RAC_CLOCK_DECL(outproj)
RAC_CLOCK_READER(outproj)
RAC_CLOCK_DECL(generator)
RAC_CLOCK_READER(generator)
struct GemmSplitArgs {
    const char *x;      // X image
    const char *w;      // W image
    float *out;         // partial mode: [slices][M][N] raw products;  affine mode: [M][ld_out] = alpha * acc + bias[n]
    const float *bias;  // affine mode only (may be null)
    float alpha;
    int M, N, K;        // K = full reduction length (multiple of 32 * slices)
    int slices;         // split-K factor (1 in affine mode)
    int tiles_x, tiles_w;   // row tiles of X, feature tiles of W
    int m16;            // ceil(M / 16)
    int tiles_per_wg;
    int affine;
    long ld_out;
};

// rows [begin, end) of X row tile i: whole 16-row MFMA tiles, as equal as possible
__device__ __forceinline__ void gs_row_tile(const GemmSplitArgs &g, int i, int &begin, int &end)
{
    begin = 16 * ((i * g.m16) / g.tiles_x);
    end = 16 * (((i + 1) * g.m16) / g.tiles_x);
    end = end < g.M ? end : g.M;
}

__global__ __launch_bounds__(768, 1) void gemm_split_kernel(const GemmSplitArgs g)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // XCD-major workgroup order (speed only: any placement computes the same tiles)
    const int nwg = gridDim.x;
    const int wg = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int ksteps = g.K / 32 / g.slices;
    const int ntiles = g.tiles_x * g.tiles_w * g.slices;
    const int t_begin = wg * g.tiles_per_wg;
    int my_tiles = ntiles - t_begin;
    my_tiles = my_tiles < g.tiles_per_wg ? my_tiles : g.tiles_per_wg;
    if (my_tiles <= 0)
        return;
    const int gsteps = my_tiles * ksteps;                          // K steps of all my tiles, one continuous sequence
    const size_t line_stride = (size_t)(g.K / 32) * GS_LINE;     // bytes between consecutive rows of an image

    if (wave >= 8) {
        // ---------------------------------------------------------------- loader waves
        // piece p of a stage = 8 lines: W rows 8p .. 8p+7 of the tile for p < 32, X rows 8(p-32) .. for p >= 32.  Lane: line
        // lane >> 3 of the eight, LDS slot lane & 7, which receives the line's 16-byte chunk slot ^ ((row >> 1) & 7).
        const int lw = wave - 8;
        const char *src[GS_PIECES_PER_LOADER];
        auto set_tile = [&](int t) {
            const int xt = t % g.tiles_x, r2 = t / g.tiles_x, wt = r2 % g.tiles_w, slice = r2 / g.tiles_w;
            int xb, xe;
            gs_row_tile(g, xt, xb, xe);
            const size_t k0 = (size_t)slice * ksteps * GS_LINE;
#pragma unroll
            for (int j = 0; j < GS_PIECES_PER_LOADER; ++j) {
                const int p = lw + 4 * j;
                const bool isx = p >= 32;
                const int line = (p & 31) * 8 + (lane >> 3);
                int r = isx ? xb + line : wt * GS_TW + line;
                const int lim = isx ? xe : g.N;
                r = r < lim ? r : lim - 1;         // lines past the edge re-read the last row; their results are not stored
                src[j] = (isx ? g.x : g.w) + (size_t)r * line_stride + k0 + (size_t)(((lane & 7) ^ ((line >> 1) & 7)) * 16);
            }
        };
        auto issue = [&](int gs) {
            const int ks = gs % ksteps;
            if (ks == 0)
                set_tile(t_begin + gs / ksteps);
            char *stage = lds + (gs % GS_STAGES) * GS_STAGE;
#pragma unroll
            for (int j = 0; j < GS_PIECES_PER_LOADER; ++j) {
                const int p = lw + 4 * j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + (size_t)ks * GS_LINE),
                                                 (__attribute__((address_space(3))) void *)(stage + p * 1024), 16, 0, 0);
            }
        };
        // prologue: steps 0 and 1 in flight; step 0 must have landed before the first barrier
        issue(0);
        if (gsteps > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        for (int gs = 0; gs < gsteps; ++gs) {
            // stage (gs+2)%3 was read in step gs-1: free since the barrier that ended it
            if (gs + 2 < gsteps) {
                issue(gs + 2);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // all but step gs+2 landed: step gs+1 is complete
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // -------------------------------------------------------------------- multiplying waves
    const int li = lane & 15, lk = lane >> 4;
    const int ww = wave >> 1, wx = wave & 1;       // 4 along W (64 features each) x 2 along X (64 rows each)
    // fragment addresses: line = tile row, chunk c (hi: lk, lo: 4 + lk) at slot c ^ ((row >> 1) & 7); MFMA tiles start at
    // multiples of 16, so (row >> 1) & 7 == (li >> 1) & 7 for every fragment row
    const int f = (li >> 1) & 7;
    const int ch = (lk ^ f) * 16, cl = ((4 + lk) ^ f) * 16;
    int w_off[4], x_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w_off[i] = (64 * ww + 16 * i + li) * GS_LINE;
        x_off[i] = GS_TW * GS_LINE + (64 * wx + 16 * i + li) * GS_LINE;
    }
    gs_f4 acc[4][4];                                // [W tile][X tile]: rows = features, cols = X rows
    __builtin_amdgcn_s_barrier();                   // step 0 has landed
    RAC_CLOCK_BEGIN();
    int gs = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int t = t_begin + ti;
        const int xt = t % g.tiles_x, r2 = t / g.tiles_x, wt = r2 % g.tiles_w, slice = r2 / g.tiles_w;
        int xb, xe;
        gs_row_tile(g, xt, xb, xe);
        // 16-row X tiles of this wave that hold real rows (wave-uniform)
        int nx = (xe - xb - 64 * wx + 15) >> 4;
        nx = nx < 0 ? 0 : (nx > 4 ? 4 : nx);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = (gs_f4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ksteps; ++ks, ++gs) {
            const char *S = lds + (gs % GS_STAGES) * GS_STAGE;
            gs_h8 wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wh[i] = *reinterpret_cast<const gs_h8 *>(S + w_off[i] + ch);
                wl[i] = *reinterpret_cast<const gs_h8 *>(S + w_off[i] + cl);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nx) {
                    const gs_h8 xh = *reinterpret_cast<const gs_h8 *>(S + x_off[j] + ch);
                    const gs_h8 xl = *reinterpret_cast<const gs_h8 *>(S + x_off[j] + cl);
                    // smallest terms first
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh, acc[i][j], 0, 0, 0);
                }
            }
            // every fragment of this stage is in registers (the MFMAs consumed them): the loaders may overwrite it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // epilogue (the loaders are already fetching the next tile): C/D layout col = li (X row), row = 4 * lk + r
        // (feature): four consecutive features per lane = one 16-byte store
        float *obase = g.affine ? g.out : g.out + (size_t)slice * g.M * g.N;
        const long ld = g.affine ? g.ld_out : (long)g.N;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = xb + 64 * wx + 16 * j + li;
            if (j < nx && row < xe) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = wt * GS_TW + 64 * ww + 16 * i + 4 * lk;
                    if (n + 3 < g.N) {
                        gs_f4 v = acc[i][j];
                        if (g.affine) {
                            const gs_f4 b = g.bias ? *reinterpret_cast<const gs_f4 *>(g.bias + n) : (gs_f4){0.f, 0.f, 0.f, 0.f};
                            v = v * g.alpha + b;
                        }
                        *reinterpret_cast<gs_f4 *>(obase + (size_t)row * ld + n) = v;
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.N)
                                obase[(size_t)row * ld + n + r] = g.affine ? acc[i][j][r] * g.alpha + (g.bias ? g.bias[n + r] : 0.f)
                                                                            : acc[i][j][r];
                    }
                }
            }
        }
    }
    RAC_CLOCK_END(outproj, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ W-stationary kernel
// parameter_generator: out[m][n] = alpha * sum_k X[m][k] W[n][k] + bias[n] with K = 256 only -- per flop the tiled kernel
// above has to take in as many operand bytes as a 32-times-longer K loop would, and the LDS-DMA intake of a CU (~32 GB/s
// with the matrix pipe busy) makes it slower than a library GEMM (151 us against 127 us).  Here the WEIGHTS never pass
// through LDS: a workgroup owns 256 features, each of its 8 waves keeps the hi and lo fragments of its 32 features for
// the whole K in registers (2 tiles x 8 K steps x 2 x 4 = 128 VGPRs, loaded once, straight from the image), and only
// the 0.9 MB X image streams through a ring of four 32-row stages (32 KB each, whole K per row, LDS-DMA, one 1 KB piece
// = one row).  Per stage and SIMD 192 MFMAs (1.5 us) stand against 32 KB of intake (22 GB/s): matrix-pipe-bound.
// All eight waves multiply AND load (four pieces each per stage; the two waves of a SIMD alternate between the LDS-DMA
// issue and their MFMA block); stores, LDS-DMA and loads share vmcnt, so the counted wait covers the stores in between.
#define GW_STAGES 4
#ifndef GW4_LDS_PAD
#define GW4_LDS_PAD 0                       /* A/B builds: extra LDS bytes requested by the four-wave shape (96 KB in all: one workgroup per CU) */
#endif
#ifndef GW_UNROLL
#define GW_UNROLL 4                         /* stages per trip of the stage loop (a multiple of GW_STAGES) */
#endif
#ifndef GW_WIDE_WAVES
#define GW_WIDE_WAVES 8                     /* waves per workgroup of the wide generator launch: 8 (one workgroup per CU) or 4 (two; A/B builds) */
#endif

struct GenArgs {
    const char *x;      // X image [M][8 lines]
    const char *w;      // W image [N][8 lines]
    const float *bias;  // [N] or null
    float *out;         // [M][ld_out]
    float alpha;
    int M, N;
    long ld_out;
    int rows_per_wg;   // rows a workgroup walks (blockIdx.y selects the chunk): M for the wide generator, a multiple of
                       // the stage height for narrow outputs, where the feature blocks alone would leave most CUs idle
};

// Two shapes of the same kernel:
//   <8, 32>  512 threads, one workgroup per CU: 256 features per workgroup, stages of 32 rows (2 x 2 MFMA tiles per wave), 128 KB ring.
//            Both waves of a SIMD belong to the same workgroup and meet at the same stage barrier: while they store a stage's tile,
//            wait for the next stage and read its first fragments, the SIMD's matrix pipe has nothing to do (round-4 counters:
//            50 % MFMA-busy).
//   <4, 16>  256 threads, TWO workgroups per CU (round 4): 128 features per workgroup, stages of 16 rows (2 x 1 tiles per wave),
//            64 KB ring each.  The two waves of a SIMD now belong to different workgroups with barriers of their own, so they drift
//            out of phase and one multiplies while the other is between stages.  Every workgroup still streams the whole X image
//            (0.9 MB, L2-resident), so the LDS-DMA intake per CU doubles (1.8 MB per launch).
// Diagnostic build only (tools/gen_phase_split.py; -DGW_STAMPS): s_memtime at four points of every stage, waves 0 and 5 of the first 256 workgroups.
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(GW_STAMPS)
#define GW_STAMP_STAGES 32
__device__ unsigned long long gw_stamp_buf[256 * 2 * GW_STAMP_STAGES * 6];
#define GW_STAMP(i)                                                                                                          \
    do {                                                                                                                     \
        if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == 5) && st < GW_STAMP_STAGES && blockIdx.x < 256 && blockIdx.y == 0) \
            gw_stamp_buf[((blockIdx.x * 2 + (wave != 0)) * GW_STAMP_STAGES + st) * 6 + (i)] =                                    \
                (i) == 5 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();                                     \
    } while (0)
extern "C" int rac_dbg_gw_stamps(unsigned long long *host_out)
{
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gw_stamp_buf), sizeof(gw_stamp_buf), 0, hipMemcpyDeviceToHost);
}
#else
#define GW_STAMP(i)
#endif

template <int WAVES, int ROWS>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) void generator_ws_kernel(const GenArgs g)
{
    constexpr int STAGE = ROWS * 1024;          // bytes: K = 256 -> 8 lines = 1 KB per row
    constexpr int J = ROWS / 16;                // 16-row MFMA tiles per stage
    constexpr int PIECES = ROWS / WAVES;        // LDS-DMA pieces (rows) per wave and stage
    static_assert(PIECES == 4 && (J == 1 || J == 2), "the counted waits below are written for 4 pieces and 2 J stores per stage");
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int n0 = blockIdx.x * (32 * WAVES) + 32 * wave; // this wave's 32 features
    const int m0 = blockIdx.y * g.rows_per_wg;            // this workgroup's rows m0 .. m0 + mrows - 1
    const int mrows = min(g.rows_per_wg, g.M - m0);
    const int nstages = (mrows + ROWS - 1) / ROWS;
    const bool full_n = n0 + 32 <= g.N;                   // every feature of this wave is stored (wave-uniform)

    // ---- X loader role: piece = one row (1 KB); wave w moves rows w, w + WAVES, ... of every stage.  LDS slot `lane` of
    // the row receives the row's 16-byte chunk lane ^ (row & 15) (source-side swizzle, see the fragment reads)
    // (ring slots are compile-time constants everywhere: with a run-time slot index hipcc cannot tell an LDS-DMA into one
    //  slot from the fragment reads of another and waits vmcnt(0) before every first read -- the pipeline collapses)
    auto issue = [&](int st, int slot) {
        char *stage = lds + slot * STAGE;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int r = wave + WAVES * j;
            int row = m0 + st * ROWS + r;
            row = row < g.M ? row : g.M - 1;           // rows past the end re-read the last row; never stored
            const char *src = g.x + (size_t)row * 1024 + (size_t)((lane ^ (r & 15)) * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(stage + r * 1024), 16, 0, 0);
        }
    };
    issue(0, 0);
    if (nstages > 1) issue(1, 1);
    if (nstages > 2) issue(2, 2);

    // ---- this wave's weights: fragment (tile t, K step ks) = W rows n0 + 16t + li, 16-byte chunk lk of the hi / lo half
    gs_h8 wh[2][8], wl[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int n = n0 + 16 * t + li;
        n = n < g.N ? n : g.N - 1;
        const char *wp = g.w + (size_t)n * 1024 + lk * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            wh[t][ks] = *reinterpret_cast<const gs_h8 *>(wp + ks * 128);
            wl[t][ks] = *reinterpret_cast<const gs_h8 *>(wp + ks * 128 + 64);
        }
    }
    gs_f4 bias4[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + 16 * t + 4 * lk;
        bias4[t] = (g.bias && n + 3 < g.N) ? *reinterpret_cast<const gs_f4 *>(g.bias + n) : (gs_f4){0.f, 0.f, 0.f, 0.f};
    }
    // (the waits below are spelled out: everything issued so far -- 12 pieces and 32 weight loads -- has to be complete)
    // (as a builtin, so that hipcc's own wait-count bookkeeping knows the weight registers have landed: behind an inline-asm wait it
    //  still placed a vmcnt(0) in front of their first use inside the stage loop -- once per trip, draining the stores of the stage before)
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment read: row r = 16j + li of the stage, chunk c = 8 ks + (hi: lk, lo: 4 + lk), LDS slot c ^ (r & 15) = c ^ li
    // A trip of the outer loop is GW_UNROLL stages with compile-time ring slots.
    RAC_CLOCK_BEGIN();
    for (int st0 = 0; st0 < nstages; st0 += GW_UNROLL) {
#pragma unroll
    for (int u = 0; u < GW_UNROLL; ++u) {
        const int st = st0 + u;
        if (st >= nstages)
            break;
        // the ring slot of stage st+3 was read in stage st-1: free since the last barrier.  In the first stage of a trip the
        // issue comes AFTER the MFMA block: at the loop's back edge hipcc cannot match the pending LDS-DMAs with the slots they
        // target and drains vmcnt before the trip's first fragment read -- with the new pieces not yet issued that costs
        // nothing (the older ones have landed), with them issued it exposed one full memory latency every four stages
        GW_STAMP(0);
        if (u != 0 && st + 3 < nstages)
            issue(st + 3, (u + 3) & (GW_STAGES - 1));
        const char *S = lds + (u & (GW_STAGES - 1)) * STAGE;
        gs_f4 acc[2][J];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < J; ++j)
                acc[t][j] = (gs_f4){0.f, 0.f, 0.f, 0.f};
        // X fragments one K step ahead of their MFMAs (two register sets): the LDS latency of step ks + 1 runs under the
        // MFMAs of step ks instead of in front of them
        gs_h8 xh[2][J], xl[2][J];
        auto frag = [&](int ks, int b) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const char *rowp = S + (16 * j + li) * 1024;
                xh[b][j] = *reinterpret_cast<const gs_h8 *>(rowp + (((8 * ks + lk) ^ li) * 16));
                xl[b][j] = *reinterpret_cast<const gs_h8 *>(rowp + (((8 * ks + 4 + lk) ^ li) * 16));
            }
        };
        frag(0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 1 < 8)
                frag(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);   // (keeps the reads above the MFMAs: hipcc otherwise sinks them to their uses)
            // product-major: independent accumulators between two MFMAs into the same one (back-to-back dependent MFMAs
            // wait out each other's latency -- the kernel ran at half the MFMA rate)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (pr == 0)
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t][ks], xh[ks & 1][j], acc[t][j], 0, 0, 0);
                        else if (pr == 1)
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][ks], xl[ks & 1][j], acc[t][j], 0, 0, 0);
                        else
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][ks], xh[ks & 1][j], acc[t][j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        GW_STAMP(1);
        if (u == 0 && st + 3 < nstages)
            issue(st + 3, 3);
        // C/D layout: col = li (X row), rows 4 lk + r = four consecutive features: one 16-byte store each
        // (Round 5, stamps of tools/gen_phase_split.py, profiles/r05_generator_phases.json: a stage is 2.58 us = MFMA block 1.25 us on the
        //  wave of a SIMD that gets the pipe first, 1.87 us on its partner (1.55 us is the pipes' own time for the two) + this epilogue
        //  0.32-0.36 + the counted wait 0.13 + the barrier.  Tried and rejected: a second accumulator set with this epilogue moved INTO
        //  the next stage's MFMA block (256 VGPRs; 87.3 us against 89.4: the block grew by what the epilogue took, a wave issues in
        //  order and its stores hold back its MFMAs), and the same with the two waves of a SIMD doing it at different K steps
        //  (3 spilled dwords: 103.6 us).)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int row = m0 + st * ROWS + 16 * j + li;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = n0 + 16 * t + 4 * lk;
                const gs_f4 v = acc[t][j] * g.alpha + bias4[t];
                if (row < m0 + mrows) {
                    if (n + 3 < g.N) {
                        *reinterpret_cast<gs_f4 *>(g.out + (size_t)row * g.ld_out + n) = v;
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.N)
                                g.out[(size_t)row * g.ld_out + n + r] = acc[t][j][r] * g.alpha + (g.bias ? g.bias[n + r] : 0.f);
                    }
                }
            }
        }
        // stage st+1 must have landed before anyone reads it.  Issue order of this wave's vector-memory operations:
        //   ... P(st+1) S(st-2) | P(st+2) S(st-1) | P(st+3) S(st)      (P: 4 pieces, S: up to 2 J stores; P before S either way)
        // A wave whose 32 features and whose stage rows are all inside the output issues exactly 2 J stores per stage: at most
        // 3 * 2J + 2 * 4 younger operations may then remain outstanding behind P(st+1) (fewer pieces follow towards the end).
        // Any other wave (features at or beyond N in the last feature block, a partial last row stage: the compiler branches
        // around stores no lane takes) issues an unknown number of stores: it may only leave the pieces themselves outstanding.
        GW_STAMP(2);
        const bool all_stores = full_n && (st + 1) * ROWS <= mrows;     // wave-uniform
        if (all_stores) {
            if (J == 2) {
                if (st + 3 < nstages)
                    asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
                else if (st + 2 < nstages)
                    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
            } else {
                if (st + 3 < nstages)
                    asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
                else if (st + 2 < nstages)
                    asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            }
        } else {
            if (st + 3 < nstages)
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (st + 2 < nstages)
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        GW_STAMP(3);
        __builtin_amdgcn_s_barrier();
        GW_STAMP(4);
        GW_STAMP(5);
    }
    }
    RAC_CLOCK_END(generator, blockIdx.y * gridDim.x + blockIdx.x);
}

// ---- weight packer: nn.Linear weight [N][K] f32 -> image [N][K/32][hi 32 | lo 32] f16 of weight * scale -------------------
__global__ __launch_bounds__(256) void gemm_split_pack_kernel(const float *__restrict__ w, _Float16 *__restrict__ img, long n4,
                                                              float scale)
{
    // one thread per 4 consecutive K values
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const rac_f4 v = rac_ld4(w + i * 4);
        rac_h4 hi, lo;
        rac_split_f16(v.x * scale, hi.x, lo.x);
        rac_split_f16(v.y * scale, hi.y, lo.y);
        rac_split_f16(v.z * scale, hi.z, lo.z);
        rac_split_f16(v.w * scale, hi.w, lo.w);
        const long e = i * 4, line = e >> 5, k = e & 31;           // rows are multiples of 32 values: lines never straddle rows
        _Float16 *dst = img + line * 64 + k;
        *reinterpret_cast<rac_h4 *>(dst) = hi;
        *reinterpret_cast<rac_h4 *>(dst + 32) = lo;
    }
}

extern "C" int rac_gemm_split_pack_fwd(const float *weight, void *image, int N, int K, float scale, void *stream)
{
    RAC_CHECK_ARG(weight && image && N >= 1 && K >= 32 && K % 32 == 0, "rac_gemm_split_pack_fwd: N=%d K=%d (K must be a multiple of 32)", N, K);
    const long n4 = (long)N * K / 4;
    const int nb = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(gemm_split_pack_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, weight,
                       reinterpret_cast<_Float16 *>(image), n4, scale);
    return rac_launch_status("rac_gemm_split_pack_fwd");
}

static int gs_launch(GemmSplitArgs &g, int tiles_per_wg, hipStream_t st, const char *what)
{
    g.tiles_x = (g.M + GS_TX - 1) / GS_TX;
    g.tiles_w = (g.N + GS_TW - 1) / GS_TW;
    g.m16 = (g.M + 15) / 16;
    g.tiles_per_wg = tiles_per_wg;
    const int ntiles = g.tiles_x * g.tiles_w * g.slices;
    if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_GEMM_SPLIT, reinterpret_cast<const void *>(gemm_split_kernel), (int)(GS_STAGES * GS_STAGE)))
        return rc_attr;
    hipLaunchKernelGGL(gemm_split_kernel, dim3((ntiles + tiles_per_wg - 1) / tiles_per_wg), dim3(768), GS_STAGES * GS_STAGE, st, g);
    return rac_launch_status(what);
}

extern "C" int rac_outproj_fwd(const void *z_image, const void *w_image, float *partials, int M, int N, int K, int slices,
                               void *stream)
{
    RAC_CHECK_ARG(z_image && w_image && partials, "rac_outproj_fwd: null pointer");
    // (N % 4 != 0 is accepted: rows then start at addresses that are not 16-byte aligned and the kernel's 16-byte stores are unaligned ones,
    //  which gfx950 global memory serves -- slower, tested at N = 130 --; the product shape has N = 256)
    RAC_CHECK_ARG(M >= 1 && N >= 1 && slices >= 1 && K >= 32 && K % (32 * slices) == 0,
                  "rac_outproj_fwd: M=%d N=%d K=%d slices=%d (K must be a multiple of 32 * slices)", M, N, K, slices);
    GemmSplitArgs g;
    g.x = reinterpret_cast<const char *>(z_image);
    g.w = reinterpret_cast<const char *>(w_image);
    g.out = partials; g.bias = nullptr; g.alpha = 1.f; g.affine = 0; g.ld_out = N;
    g.M = M; g.N = N; g.K = K; g.slices = slices;
    return gs_launch(g, 1, (hipStream_t)stream, "rac_outproj_fwd");
}

extern "C" int rac_generator_fwd(const void *x_image, const void *w_image, const float *bias, float alpha, float *out, int64_t ld_out,
                                 int M, int N, int K, void *stream)
{
    RAC_CHECK_ARG(x_image && w_image && out, "rac_generator_fwd: null pointer");
    RAC_CHECK_ARG(M >= 1 && N >= 1 && K >= 32 && K % 32 == 0 && ld_out >= N && ld_out % 4 == 0 && (N % 4 == 0 || K == 256),
                  "rac_generator_fwd: M=%d N=%d K=%d ld_out=%lld (K %% 32 and ld_out %% 4 must be 0; N %% 4 too unless K == 256)", M, N, K,
                  (long long)ld_out);
    if (K == 256) {
        GenArgs a;
        a.x = reinterpret_cast<const char *>(x_image);
        a.w = reinterpret_cast<const char *>(w_image);
        a.bias = bias; a.out = out; a.alpha = alpha; a.M = M; a.N = N; a.ld_out = ld_out;
        if (GW_WIDE_WAVES == 4 && N >= 128 * 128) {
            // the wide generator (65536 features): 512 workgroups of four waves, two per CU, every one walks all rows
            a.rows_per_wg = (M + 15) / 16 * 16;
            if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_GENERATOR4, reinterpret_cast<const void *>(generator_ws_kernel<4, 16>), GW_STAGES * 16 * 1024 + GW4_LDS_PAD))
                return rc_attr;
            hipLaunchKernelGGL((generator_ws_kernel<4, 16>), dim3((N + 127) / 128, 1), dim3(256), GW_STAGES * 16 * 1024 + GW4_LDS_PAD, (hipStream_t)stream, a);
            return rac_launch_status("rac_generator_fwd");
        }
        // narrow outputs (the 2189 features of the sampling Linears, not the generator's 65536): the rows are cut into chunks so that
        // feature blocks x chunks covers the CUs; every chunk re-reads its weight rows (L2) for a few row stages of work, so the
        // weight prologue is most of a workgroup's life.  Round 5: the FOUR-wave shape here (128 features = 128 KB of weights per
        // workgroup instead of 256 KB, 16-row stages; about 256 workgroups): 15.6 -> 12.3 us at N = 2189, M = 900
        // (the same shape LOSES on the wide generator, 104-106 us against 88-90: there the stage loop is everything).
#if !defined(GW_NARROW_WAVES) || GW_NARROW_WAVES == 4
        if (N < 128 * 128) {
            const int fb = (N + 127) / 128;
            int ch = fb >= 256 ? 1 : 256 / fb;      // about one workgroup per CU (rows per workgroup 32 / 48 / 64 / 80 / 112: 15.7 / 13.9 / 12.9 / 12.3 / 12.7 us)
            const int maxc = (M + 15) / 16;
            ch = ch > maxc ? maxc : ch;
            a.rows_per_wg = ((M + ch - 1) / ch + 15) / 16 * 16;
#ifdef GW_NARROW_ROWS
            a.rows_per_wg = GW_NARROW_ROWS;      /* A/B builds */
#endif
            ch = (M + a.rows_per_wg - 1) / a.rows_per_wg;
            if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_GENERATOR4, reinterpret_cast<const void *>(generator_ws_kernel<4, 16>), GW_STAGES * 16 * 1024 + GW4_LDS_PAD))
                return rc_attr;
            hipLaunchKernelGGL((generator_ws_kernel<4, 16>), dim3(fb, ch), dim3(256), GW_STAGES * 16 * 1024 + GW4_LDS_PAD, (hipStream_t)stream, a);
            return rac_launch_status("rac_generator_fwd");
        }
#endif
        // the eight-wave shape for narrow outputs (rounds 2-4; A/B builds with -DGW_NARROW_WAVES=8): one workgroup per CU (128 KB of
        // LDS), so at most 256 workgroups: a 257th would wait for a whole round.  (Measured at N = 2189, M = 900: 135 workgroups
        // 14.8 us, 261 workgroups 20.5 us; an XCD-major item order that keeps a feature block's chunks on one L2 changed nothing.)
        constexpr int ROWS = 32;
        const int fblocks = (N + 255) / 256;
        int chunks = fblocks >= 128 ? 1 : 256 / fblocks;
        const int max_chunks = (M + ROWS - 1) / ROWS;
        chunks = chunks > max_chunks ? max_chunks : chunks;
        a.rows_per_wg = ((M + chunks - 1) / chunks + ROWS - 1) / ROWS * ROWS;
        chunks = (M + a.rows_per_wg - 1) / a.rows_per_wg;
        if (const int rc_attr = rac_set_dynamic_lds_once(RAC_ATTR_GENERATOR, reinterpret_cast<const void *>(generator_ws_kernel<8, 32>), GW_STAGES * ROWS * 1024))
            return rc_attr;
        hipLaunchKernelGGL((generator_ws_kernel<8, 32>), dim3(fblocks, chunks), dim3(512), GW_STAGES * ROWS * 1024, (hipStream_t)stream, a);
        return rac_launch_status("rac_generator_fwd");
    }
    GemmSplitArgs g;
    g.x = reinterpret_cast<const char *>(x_image);
    g.w = reinterpret_cast<const char *>(w_image);
    g.out = out; g.bias = bias; g.alpha = alpha; g.affine = 1; g.ld_out = ld_out;
    g.M = M; g.N = N; g.K = K; g.slices = 1;
    // other K: the tiled kernel; one workgroup per W tile walks all row tiles (its W rows stay in its XCD's L2)
    return gs_launch(g, (M + GS_TX - 1) / GS_TX, (hipStream_t)stream, "rac_generator_fwd");
}
