// gemm_split.hip -- split-precision GEMMs of AdaptiveMixing on the f16 matrix cores (gfx950), hand-written:
//   rac_outproj_fwd : out_proj, nn.Linear(32768 -> 256) over 900 queries (models/racformer_transformer.py:566,606), as
//                     a split-K GEMM  partial[s] = Z[:, slice s] @ W[:, slice s]^T
//
// Arithmetic (same as conv3x3.hip): every operand is v * 2^e = hi + lo (two f16, 22 significant bits); the three
// leading products lo*hi + hi*lo + hi*hi are accumulated in fp32 by v_mfma_f32_16x16x32_f16 -- the dropped lo*lo term
// is 2^-22 relative, the result matches an fp32 GEMM to fp32 rounding.  Both powers of two are undone by the consumer
// (rac_add_ln_fwd's a_scale), so partials leave the kernel unscaled.
//
// Operand images (one format for A and B): row r, per 32 values of K one 128-byte line [hi 32 | lo 32] f16.
//   Z image  [M][K/32][hi 32 | lo 32]  written by rac_mixing_fwd (out_split)          118 MB for 900 x 32768
//   W image  [N][K/32][hi 32 | lo 32]  packed once from the nn.Linear weight [N][K]   33.5 MB
// Compared with the K-concatenated [hi | hi | lo] x [hi | lo | hi] operands a library GEMM needs, each value is stored
// once (4 B instead of 6 B): 59 MB less written by the mixing kernel and 76 MB less read here, per layer.
//
// Workgroup = 512 threads: 4 multiplying waves (2 x 2, 64 x 64 = 4 x 4 MFMA tiles each) + 4 loader waves; tile = 128 rows x
// 128 columns x one K slice.  K loop in steps of 32: per step 128 + 128 lines of 128 B = 32 KB go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of four stages (128 KB), three steps ahead of
// the MFMAs (counted vmcnt, one barrier per step).  LDS-DMA writes a wave-instruction's 1 KB linearly (8 lines), so bank
// conflicts are avoided on the SOURCE side: LDS slot p of line r receives the line's 16-byte chunk p ^ ((r >> 1) & 7),
// and the fragment reads apply the same XOR -- every ds_read_b128 is conflict-free.
// Grid = row tiles x column tiles x K slices = 8 x 2 x 16 = 256 workgroups for out_proj: one per CU.
#include "rac_common.h"

typedef _Float16 gs_h8 __attribute__((ext_vector_type(8)));
typedef float gs_f4 __attribute__((ext_vector_type(4)));

#define GS_TM 128
#define GS_TN 128
#define GS_LINE 128                        /* bytes per row and K step of 32: [hi 32 | lo 32] f16 */
#define GS_STAGE (2 * GS_TM * GS_LINE)     /* 32 KB: A tile then B tile */
#define GS_STAGES 4
#define GS_PIECES 32                       /* 1 KB LDS-DMA pieces per stage: 16 for A, 16 for B; 8 per loader wave */

struct GemmSplitArgs {
    const char *a;    // A image
    const char *b;    // B image
    float *out;       // [S][M][N] partial products (unscaled)
    int M, N, K;      // K = full reduction length (multiple of 32 * slices)
    int slices;       // split-K factor; blockIdx.z
};

// One 128 x 128 x (K / slices) tile.  512 threads: waves 0-3 multiply (2 x 2, 64 x 64 each), waves 4-7 only load.
//   loader wave, step s : issue the 8 pieces of K step s+3 into ring stage (s+3)%4, wait until all but the youngest 16
//                         pieces (steps s+2, s+3) have landed -> step s+1 is complete, barrier
//   compute wave, step s: 16 fragment reads + 48 MFMAs on stage s%4, wait for its own LDS reads, barrier
// The barrier at the end of step s therefore publishes stage (s+1)%4 to the readers and frees stage s%4 for the loaders
// (they write it in step s+1).  The multiplying waves never issue a vector-memory instruction: an LDS-DMA piece costs its
// wave 60-185 issue cycles (MI355X_MICROARCH.md), 16 of them per K step in front of the MFMAs made the first version of
// this kernel run at 2.2 us per 64-deep step against 0.7 us of MFMA time.
__global__ __launch_bounds__(512, 1) void gemm_split_kernel(const GemmSplitArgs g)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int row0 = blockIdx.x * GS_TM, col0 = blockIdx.y * GS_TN;
    const int ksteps = g.K / 32 / g.slices;
    const size_t line_stride = (size_t)(g.K / 32) * GS_LINE;     // bytes between consecutive rows of an image
    const size_t k0 = (size_t)blockIdx.z * ksteps * GS_LINE;      // byte offset of this slice inside a row

    if (wave >= 4) {
        // ---------------------------------------------------------------- loader waves
        // piece p of a stage = 8 lines (rows 8p .. 8p+7 of the A tile for p < 16, of the B tile for p >= 16); lane: line
        // lane >> 3, LDS slot lane & 7, which receives the line's 16-byte chunk slot ^ ((row >> 1) & 7)
        const int lw = wave - 4;
        const char *src[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = lw + 4 * j;
            const bool isb = p >= 16;
            const int line = (p & 15) * 8 + (lane >> 3);
            int r = (isb ? col0 : row0) + line;
            const int lim = isb ? g.N : g.M;
            r = r < lim ? r : lim - 1;             // lines past the edge re-read the last row; their results are not stored
            src[j] = (isb ? g.b : g.a) + (size_t)r * line_stride + k0 + (size_t)(((lane & 7) ^ ((line >> 1) & 7)) * 16);
        }
        auto issue = [&](int ks) {
            char *stage = lds + (ks & (GS_STAGES - 1)) * GS_STAGE;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = lw + 4 * j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + (size_t)ks * GS_LINE),
                                                 (__attribute__((address_space(3))) void *)(stage + p * 1024), 16, 0, 0);
            }
        };
        // prologue: steps 0, 1, 2 in flight; step 0 must have landed before the first barrier
        issue(0);
        if (ksteps > 1) issue(1);
        if (ksteps > 2) issue(2);
        if (ksteps > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ksteps > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int ks = 0; ks < ksteps; ++ks) {
            if (ks + 3 < ksteps) {
                issue(ks + 3);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // steps ks+2, ks+3 may still be in flight
            } else if (ks + 2 < ksteps) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // only step ks+2 behind ks+1
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // -------------------------------------------------------------------- multiplying waves
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    gs_f4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            acc[m][n] = (gs_f4){0.f, 0.f, 0.f, 0.f};
    // fragment addresses: line = tile row, chunk c (hi: lk, lo: 4 + lk) at slot c ^ ((row >> 1) & 7); tiles start at
    // multiples of 16, so (row >> 1) & 7 == (li >> 1) & 7 for every fragment row
    const int f = (li >> 1) & 7;
    const int ch = (lk ^ f) * 16, cl = ((4 + lk) ^ f) * 16;
    int a_off[4], b_off[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        a_off[m] = (64 * wm + 16 * m + li) * GS_LINE;
        b_off[m] = GS_TM * GS_LINE + (64 * wn + 16 * m + li) * GS_LINE;
    }
    __builtin_amdgcn_s_barrier();          // step 0 has landed
    for (int ks = 0; ks < ksteps; ++ks) {
        const char *S = lds + (ks & (GS_STAGES - 1)) * GS_STAGE;
        gs_h8 bh[4], bl[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bh[n] = *reinterpret_cast<const gs_h8 *>(S + b_off[n] + ch);
            bl[n] = *reinterpret_cast<const gs_h8 *>(S + b_off[n] + cl);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const gs_h8 ah = *reinterpret_cast<const gs_h8 *>(S + a_off[m] + ch);
            const gs_h8 al = *reinterpret_cast<const gs_h8 *>(S + a_off[m] + cl);
            // smallest terms first
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], acc[m][n], 0, 0, 0);
        }
        // every fragment of this stage is in registers (the MFMAs above consumed them): the loaders may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: C/D layout col = li, row = 4 * lk + r within a 16 x 16 tile
    float *obase = g.out + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 64 * wm + 16 * m + 4 * lk + r;
            if (row < g.M) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int col = col0 + 64 * wn + 16 * n + li;
                    if (col < g.N)
                        obase[(size_t)row * g.N + col] = acc[m][n][r];
                }
            }
        }
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

extern "C" int rac_outproj_fwd(const void *z_image, const void *w_image, float *partials, int M, int N, int K, int slices,
                               void *stream)
{
    RAC_CHECK_ARG(z_image && w_image && partials, "rac_outproj_fwd: null pointer");
    RAC_CHECK_ARG(M >= 1 && N >= 1 && slices >= 1 && K >= 32 && K % (32 * slices) == 0,
                  "rac_outproj_fwd: M=%d N=%d K=%d slices=%d (K must be a multiple of 32 * slices)", M, N, K, slices);
    GemmSplitArgs g;
    g.a = reinterpret_cast<const char *>(z_image);
    g.b = reinterpret_cast<const char *>(w_image);
    g.out = partials;
    g.M = M; g.N = N; g.K = K; g.slices = slices;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  GS_STAGES * GS_STAGE);
        attr_set = true;
    }
    const dim3 grid((M + GS_TM - 1) / GS_TM, (N + GS_TN - 1) / GS_TN, slices);
    hipLaunchKernelGGL(gemm_split_kernel, grid, dim3(512), GS_STAGES * GS_STAGE, (hipStream_t)stream, g);
    return rac_launch_status("rac_outproj_fwd");
}
