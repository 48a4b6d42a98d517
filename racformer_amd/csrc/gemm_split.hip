// gemm_split.hip -- split-precision GEMMs of AdaptiveMixing on the f16 matrix cores (gfx950), hand-written:
//   rac_outproj_fwd : out_proj, nn.Linear(32768 -> 256) over 900 queries (models/racformer_transformer.py:566,606), as
//                     a split-K GEMM  partial[s] = Z[:, slice s] @ W[:, slice s]^T
//
// Arithmetic (same as conv3x3.hip): every operand is v * 2^e = hi + lo (two f16, 22 significant bits); the three
// leading products lo*hi + hi*lo + hi*hi are accumulated in fp32 by v_mfma_f32_16x16x32_f16 -- the dropped lo*lo term
// is 2^-22 relative, the result matches an fp32 GEMM to fp32 rounding.  Both powers of two are undone by the consumer
// (rac_add_ln_fwd's a_scale), so partials leave the kernel unscaled.
//
// Operand images (one format for A and B): row r, per 64 values of K one 256-byte line [hi 64 | lo 64] f16.
//   Z image  [M][K/64][hi 64 | lo 64]  written by rac_mixing_fwd (out_image)          118 MB for 900 x 32768
//   W image  [N][K/64][hi 64 | lo 64]  packed once from the nn.Linear weight [N][K]   33.5 MB
// Compared with the K-concatenated [hi | hi | lo] x [hi | lo | hi] operands a library GEMM needs, each value is stored
// once (4 B instead of 6 B): 59 MB less written by the mixing kernel and 76 MB less read here, per layer.
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile = 128 rows x 128 columns x one K slice; wave = 64 x 64 = 4 x 4 MFMA
// tiles.  K loop in steps of 64: per step 128 + 128 lines of 256 B = 64 KB go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write), two stages = 128 KB; the loads of step k+1 are issued
// before the 96 MFMAs per wave of step k and waited for (vmcnt(0)) after them.  LDS-DMA writes a wave-instruction's
// 1 KB linearly (4 lines), so bank conflicts are avoided on the SOURCE side: LDS slot p of line r receives the line's
// 16-byte chunk p ^ (r & 15), and the fragment reads apply the same XOR -- every ds_read_b128 is conflict-free.
// Grid = row tiles x column tiles x K slices = 8 x 2 x 16 = 256 workgroups for out_proj: one per CU.
#include "rac_common.h"

typedef _Float16 gs_h8 __attribute__((ext_vector_type(8)));
typedef float gs_f4 __attribute__((ext_vector_type(4)));

#define GS_TM 128
#define GS_TN 128
#define GS_LINE 256                        /* bytes per row and K step */
#define GS_STAGE (2 * GS_TM * GS_LINE)     /* 64 KB: A tile then B tile */

struct GemmSplitArgs {
    const char *a;    // A image
    const char *b;    // B image
    float *out;       // [S][M][N] partial products (unscaled)
    int M, N, K;      // K = full reduction length (multiple of 64 * slices)
    int slices;       // split-K factor; blockIdx.z
};

__device__ __forceinline__ void gs_wait_all_and_sync()
{
    // LDS-DMA counts on vmcnt; a plain __syncthreads() would do the same wait, spelled out here because the loads of the
    // next stage are deliberately in flight across the MFMA block and must have landed before any wave reads them
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0) expcnt(0)
    __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(256, 1) void gemm_split_kernel(const GemmSplitArgs g)
{
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * GS_TM, col0 = blockIdx.y * GS_TN;
    const int ksteps = g.K / 64 / g.slices;
    const size_t line_stride = (size_t)(g.K / 64) * GS_LINE;     // bytes between consecutive rows of an image
    const size_t k0 = (size_t)blockIdx.z * ksteps * GS_LINE;      // byte offset of this slice inside a row

    // staging role: a wave-instruction moves 4 lines (1 KB); wave w issues instructions w, w+4, ... of the 64 per step
    // (32 for the A tile, 32 for the B tile).  Lane: line (lane >> 4) of the four, LDS slot p = lane & 15, source chunk
    // p ^ (line & 15).
    const char *src[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int instr = wave + 4 * j;            // 0..63
        const bool isb = instr >= 32;
        const int line = (instr & 31) * 4 + (lane >> 4);   // 0..127 within the tile
        int r = (isb ? col0 : row0) + line;
        const int lim = isb ? g.N : g.M;
        r = r < lim ? r : lim - 1;                 // lines past the edge re-read the last row; their results are not stored
        src[j] = (isb ? g.b : g.a) + (size_t)r * line_stride + k0 + (size_t)(((lane & 15) ^ (line & 15)) * 16);
    }
    auto issue = [&](int ks, int stage) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int instr = wave + 4 * j;
            // wave-uniform LDS base of this instruction's 1 KB; the hardware adds lane * 16
            char *dst = lds + stage * GS_STAGE + instr * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + (size_t)ks * GS_LINE),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    gs_f4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            acc[m][n] = (gs_f4){0.f, 0.f, 0.f, 0.f};

    // fragment addresses: line = tile row, chunk c (hi: 4*sub + lk, lo: 8 + 4*sub + lk) at slot c ^ (line & 15)
    int a_off[4], b_off[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        a_off[m] = (64 * wm + 16 * m + li) * GS_LINE;
        b_off[m] = GS_TM * GS_LINE + (64 * wn + 16 * m + li) * GS_LINE;
    }
    const int sw = li;   // (line & 15) == li for every fragment row: tiles start at multiples of 16

    issue(0, 0);
    gs_wait_all_and_sync();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int st = ks & 1;
        if (ks + 1 < ksteps)
            issue(ks + 1, st ^ 1);
        const char *S = lds + st * GS_STAGE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ch = ((4 * sub + lk) ^ sw) * 16, cl = ((8 + 4 * sub + lk) ^ sw) * 16;
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
        }
        gs_wait_all_and_sync();
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

// ---- weight packer: nn.Linear weight [N][K] f32 -> image [N][K/64][hi 64 | lo 64] f16 of weight * scale -------------------
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
        const long e = i * 4, line = e >> 6, k = e & 63;           // rows are multiples of 64 values: lines never straddle rows
        _Float16 *dst = img + line * 128 + k;
        *reinterpret_cast<rac_h4 *>(dst) = hi;
        *reinterpret_cast<rac_h4 *>(dst + 64) = lo;
    }
}

extern "C" int rac_gemm_split_pack_fwd(const float *weight, void *image, int N, int K, float scale, void *stream)
{
    RAC_CHECK_ARG(weight && image && N >= 1 && K >= 64 && K % 64 == 0, "rac_gemm_split_pack_fwd: N=%d K=%d (K must be a multiple of 64)", N, K);
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
    RAC_CHECK_ARG(M >= 1 && N >= 1 && slices >= 1 && K >= 64 && K % (64 * slices) == 0,
                  "rac_outproj_fwd: M=%d N=%d K=%d slices=%d (K must be a multiple of 64 * slices)", M, N, K, slices);
    GemmSplitArgs g;
    g.a = reinterpret_cast<const char *>(z_image);
    g.b = reinterpret_cast<const char *>(w_image);
    g.out = partials;
    g.M = M; g.N = N; g.K = K; g.slices = slices;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * GS_STAGE);
        attr_set = true;
    }
    const dim3 grid((M + GS_TM - 1) / GS_TM, (N + GS_TN - 1) / GS_TN, slices);
    hipLaunchKernelGGL(gemm_split_kernel, grid, dim3(256), 2 * GS_STAGE, (hipStream_t)stream, g);
    return rac_launch_status("rac_outproj_fwd");
}
