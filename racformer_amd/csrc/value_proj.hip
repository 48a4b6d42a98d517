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
// generator_ws_kernel) and walks its pixel blocks in stages of 32 pixels.  Per stage, wave w also takes in channels
// 32w .. 32w+31 of the 32 pixels (lane = pixel quad x channel quad: 4 x float4), the per-pixel maxima are combined across
// the waves through LDS, and the scaled hi / lo values go to LDS in the fragment layout of gemm_split.hip (1 KB per pixel,
// 16-byte chunk c at slot c ^ (pixel & 15)).
// Round 5: the pixels come through an LDS-DMA ring of raw fp32 stages (buffer_load ... lds, VP_RING stages of 32 KB, each wave its own
// 4 KB of a stage: no barrier between a wave's DMA and its own reads), issued VP_RING stages ahead.  Rounds 2-4 loaded the next stage
// into registers under the MFMAs: with 252 VGPRs there was room for ONE stage.
// Where the time is (profiles/r05_value_proj_phases.json, tools/vp_phase_split.py): the wait for a stage's pixels is ZERO -- not an HBM
// wait.  A stage is 4.0 us: raw rows + in-wave maxima 0.4, exchange of the maxima + hi / lo image 0.6-0.8, piece issue 0.2, MFMAs
// 1.0-1.7 (the matrix pipes need 1.48 for the two waves of a SIMD), epilogue 0.3, two barriers; every phase but the MFMAs is a chain
// of LDS round trips that two waves per SIMD cannot hide (the weights fill the register file: no third wave).  Ablations: without
// MFMAs 91 us, without stores 84, without ANY pixel load 83 of 98 (cold caches).  Building image s + 1 under the MFMAs of stage s,
// with the two waves of a SIMD in opposite order, was measured and rejected (profiles/r05_value_proj_phases_pipelined_rejected.json:
// 102 us, every side took as long beside the other as alone).
#include "rac_common.h"

typedef _Float16 vp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 vp_h4 __attribute__((ext_vector_type(4)));
typedef float vp_f4 __attribute__((ext_vector_type(4)));

#define VP_ROWS 32
#ifndef VP_RING
#define VP_RING 3                                /* raw stages in the LDS-DMA ring = stages in flight */
#endif
#define VP_BUF (VP_ROWS * 1024 + VP_ROWS * 4)   /* X image of a stage + its per-row output scales */
#define VP_RAW (VP_ROWS * 256 * 4)              /* one raw stage: [256 channels][32 pixels] fp32 */
#define VP_RAW0 (VP_BUF + 8 * VP_ROWS * 4)      /* X image (one buffer: barrier (A) of a stage comes after every wave's MFMAs of the one before), partial maxima, then the ring */
#define VP_LDS (VP_RAW0 + VP_RING * VP_RAW)
#define VP_LDS_Q16 (VP_LDS + 8 * VP_ROWS * 4)   /* + the waves' partial block maxima of the q16 epilogue */

// Diagnostic build only (tools/vp_phase_split.py; -DVP_STAMPS): s_memtime at the phase boundaries of every stage, waves 0 and 5 of
// every workgroup, collected in LDS and copied out at the end of the kernel.  Not part of the product library or its ABI.
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_STAMPS)
#define VP_STAMP_STAGES 32
#define VP_STAMP_N (2 * VP_STAMP_STAGES * 10)
#define VP_STAMP_LDS (VP_STAMP_N * 8)
__device__ unsigned long long vp_stamp_buf[256 * VP_STAMP_N];
#define VP_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (lane == 0 && (wave == 0 || wave == 5) && vp_nst < VP_STAMP_STAGES)                                   \
            vp_stamps[((wave != 0) * VP_STAMP_STAGES + vp_nst) * 10 + (i)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define VP_STAMP_LDS 0
#define VP_STAMP(i)
#endif

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
    int frames;          // F
    int blocks;          // HW / VP_ROWS pixel blocks per frame: workgroup b takes blocks b, b + gridDim.x, ... through all frames
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
// The workgroup barrier of the stage loop.  __syncthreads() is a workgroup-scope release fence + s_barrier + acquire fence, and for the
// fence hipcc waits for every LDS-DMA in flight (the pieces are LDS writes another wave might read): in every stage but the loop
// header's that was a vmcnt(4) in front of barrier (A) -- the ring was drained to the stage fetched last before the stage in hand had
// even been scaled.  Here no wave ever reads another wave's pieces (a wave's raw rows are its own; what crosses waves are the maxima and
// the X image, plain LDS stores): lgkmcnt(0) is the whole requirement.  The wavefront-scope fences emit nothing and keep the compiler
// from moving LDS accesses across the barrier.
__device__ __forceinline__ void vp_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0); vmcnt 63, expcnt 7: untouched
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool QOUT, bool HAS_ADD>
__device__ __forceinline__ void value_proj_body(const ValueProjArgs &g, char *__restrict__ lds, char *__restrict__ ring,
                                                unsigned *__restrict__ pmax)
{
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int pq = lane & 7, cs = lane >> 3;        // loader role: pixels 4pq .. 4pq+3, channels 32 wave + 4cs .. +3
    // Stage order: workgroup b takes the pixel blocks b, b + G, b + 2G, ... (32 pixels each) and walks every block through ALL
    // frames before the next one.  (a) The additive term belongs to the pixel, not the frame: its 32 x 256 tile is loaded once per
    // block and stays in the 16 registers it already had -- read per stage it was as many bytes as the maps themselves (134 MB per
    // stream at 8 frames, out of L2 / the Infinity Cache).  (b) At any moment the G workgroups are at neighbouring blocks of one
    // frame: the 128-byte pieces of a channel row that are in flight together are neighbours in memory (32 KB runs per channel)
    // instead of 2 KB apart.  (Round 4 and the first half of round 5 gave a workgroup 512 consecutive pixels of one frame.)
    // (pixel indices are 32-bit: the launcher checks the maps stay below 4 GiB)
    const int G = (int)gridDim.x, b0 = (int)blockIdx.x, F = g.frames;
    if (b0 >= g.blocks)
        return;
    const int nblk = (g.blocks - b0 + G - 1) / G;
    const int nstages = nblk * F;
    float *smax = reinterpret_cast<float *>(lds + VP_BUF);              // [8 waves][32 pixels]
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_STAMPS)
    unsigned long long *vp_stamps = reinterpret_cast<unsigned long long *>(lds + VP_LDS_Q16);
    int vp_nst = 0;
    for (int i = tid; i < VP_STAMP_N; i += 512)
        vp_stamps[i] = 0;
#endif

    // raw stage -> ring slot: this wave's 32 channel rows (128 B each) as four 1 KB LDS-DMA pieces of 8 rows.  Piece position q
    // (0..7) receives channel row q ^ ((q >> 2) & 1): rows r and r + 4 -- which a quarter-wave of the readers below touches together --
    // land at different parities of the 128-byte row grid (different halves of the 64 banks).
    const int dq = lane >> 3, drow = dq ^ ((dq >> 2) & 1), dslot = lane & 7;
    // (buffer_load ... lds over a descriptor of the maps, not global_load_lds: hipcc books the global form as a FLAT access that may
    //  return out of order and turns every later vector-memory wait into vmcnt(0) -- the epilogue's wait for the additive term would
    //  drain the ring each stage; the buffer form gets counted waits)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.x), 0, g.x_bytes, 0x00020000);
    // the fetch cursor runs VP_RING stages ahead of the stage in hand (past the end it stays on the last stage: fetched again, never
    // read -- every stage issues its four pieces, the counted waits assume it)
    int pf_blk = b0, pf_f = 0, pf_n = 0;
    auto dma = [&](int slot) {
        const int p = pf_blk * VP_ROWS + 4 * dslot;
        const unsigned voff = (unsigned)(((pf_f * 256 + 32 * wave + drow) * g.HW + p) * 4);
        char *dst = ring + slot * VP_RAW + wave * 4096;
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_EXP_NODMA)
        if (g.HW < 0)
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(dst + i * 1024), 16,
                                                     voff + (unsigned)(8 * i * g.HW * 4), 0, 0, 0);
        if (pf_n + 1 < nstages) {
            ++pf_n;
            if (++pf_f == F) {
                pf_f = 0;
                pf_blk += G;
            }
        }
    };
    // reader role: channel rows 4cs .. 4cs+3 of the wave's 32, pixels 4pq .. 4pq+3
    unsigned roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * cs + i, q = r & 7;
        roff[i] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)ring +
                  (unsigned)(wave * 4096 + (r >> 3) * 1024 + (q ^ ((q >> 2) & 1)) * 128 + pq * 16);
    }
#pragma unroll
    for (int s0 = 0; s0 < VP_RING; ++s0)
        dma(s0);

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
    // the additive term of the block in hand (HAS_ADD: reloaded by the first stage of every block; else the bias, once)
    vp_f4 addv[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const vp_f4 b4 = (g.bias && !HAS_ADD) ? *reinterpret_cast<const vp_f4 *>(g.bias + 32 * wave + 16 * t + 4 * lk) : (vp_f4){0.f, 0.f, 0.f, 0.f};
        addv[0][t] = b4;
        addv[1][t] = b4;
    }

    // Everything issued so far (the 32 weight fragments, the first stages' pixels) has to have LANDED before the stage loop is entered,
    // and the compiler has to know it: hipcc places the vmcnt waits for the weight registers at their first uses INSIDE the loop
    // (vmcnt(30) ... vmcnt(0) between the MFMAs), and from the second stage on those same waits drain the loads of the NEXT
    // stage that were issued just above them -- the prefetch ran inside the MFMA phase instead of under it (round-4 ISA reading:
    // 63 % of the wave cycles parked in s_waitcnt).  simm16 = vmcnt(0), expcnt / lgkmcnt untouched.
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // FIRST: the first stage of a block (loads the block's additive tile).  Two copies of the stage rather than a branch around the
    // four loads: with a branch the number of loads in flight is path-dependent and every later counted wait falls back to the
    // conservative path.
    auto stage = [&](auto FIRST, const int blk, const int f, const int slot) {
        char *B = lds;
        float *salpha = reinterpret_cast<float *>(B + VP_ROWS * 1024);
        VP_STAMP(0);
        // This wave's rows of the stage have landed: everything issued behind them may still be in flight -- the stage that fetched
        // them issued 4 stores after the pieces, each of the VP_RING - 1 stages since 4 pieces + 4 stores.  (A block's first stage
        // adds four loads of the additive tile, the q16 epilogue of the even waves two scale stores: more operations behind the
        // pieces than counted here, i.e. the wait is satisfied a little later than it had to be -- never earlier.)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (VP_RING - 1) + 4) : "memory");
        VP_STAMP(1);
        // The four reads of the raw stage are inline assembly: as ordinary LDS loads hipcc cannot tell them from the OTHER slots' pending
        // LDS-DMA (run-time offsets into one LDS array: SIInsertWaitcnts then waits for every LDS-DMA in flight, i.e. for the stages
        // that have just been prefetched).  The counted wait above is what orders them behind their own stage's pieces.
        vp_f4 xv[4];
        {
            const unsigned sb = (unsigned)(slot * VP_RAW);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3])
                         : "v"(sb + roff[0]), "v"(sb + roff[1]), "v"(sb + roff[2]), "v"(sb + roff[3]));
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
        VP_STAMP(2);
        vp_barrier();                                     // (A) partial maxima visible; everyone is past the MFMAs of the stage before
        VP_STAMP(3);
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
        VP_STAMP(4);
        vp_barrier();                                     // (B) the stage's X image is complete
        VP_STAMP(5);
        // the block's additive tile first (first stage of a block only), then the pixels VP_RING stages ahead into the slot this
        // stage has just been read out of: both in flight under the MFMAs, and the epilogue's wait for the (older) additive term
        // leaves the newest pieces outstanding
        const int p0 = blk * VP_ROWS;
        const int gp = f * g.HW + p0;
        if constexpr (HAS_ADD && FIRST.value) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    addv[j][t] = *reinterpret_cast<const vp_f4 *>(g.add + (size_t)(p0 + 16 * j + li) * 256 + 32 * wave + 16 * t + 4 * lk);
        }
        dma(slot);
        __builtin_amdgcn_sched_barrier(0);
        VP_STAMP(6);

        vp_f4 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[t][j] = (vp_f4){0.f, 0.f, 0.f, 0.f};
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_EXP_NOMFMA)
        if (g.HW < 0)
#endif
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
        __builtin_amdgcn_sched_barrier(0);
        VP_STAMP(7);
        // ---- epilogue: C/D layout col = li (pixel), rows 4 lk + r = four consecutive features: one 16-byte store
        if constexpr (QOUT) {
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
            vp_barrier();                                 // (C) the partner wave's half-block maxima are visible
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
        } else {
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
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_EXP_NOSTORE)
                    if (acc[t][j][0] == 123.456f)
#endif
                    *reinterpret_cast<vp_f4 *>(g.out + (size_t)(gp + r) * 256 + n) = __builtin_elementwise_fma(acc[t][j], (vp_f4){a, a, a, a}, addv[j][t]);
                }
            }
        }
        VP_STAMP(8);
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_STAMPS)
        if (lane == 0 && (wave == 0 || wave == 5) && vp_nst < VP_STAMP_STAGES)
            vp_stamps[((wave != 0) * VP_STAMP_STAGES + vp_nst) * 10 + 9] = __builtin_amdgcn_s_memrealtime();
        ++vp_nst;
#endif
    };
    int slot = 0;
    auto next = [&]() { slot = slot + 1 == VP_RING ? 0 : slot + 1; };
    for (int k = 0, blk = b0; k < nblk; ++k, blk += G) {
        if constexpr (HAS_ADD) {
            stage(rac_ic<1>{}, blk, 0, slot);
            next();
            for (int f = 1; f < F; ++f, next())
                stage(rac_ic<0>{}, blk, f, slot);
        } else {
            for (int f = 0; f < F; ++f, next())
                stage(rac_ic<0>{}, blk, f, slot);
        }
    }
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_STAMPS)
    __syncthreads();
    for (int i = tid; i < VP_STAMP_N; i += 512)
        vp_stamp_buf[(size_t)blockIdx.x * VP_STAMP_N + i] = vp_stamps[i];
#endif
}

#if defined(RAC_DIAGNOSTIC_BUILD) && defined(VP_STAMPS)
extern "C" int rac_dbg_vp_stamps(unsigned long long *host_out, int n_wgs)
{
    if (n_wgs > 256)
        n_wgs = 256;
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vp_stamp_buf), sizeof(unsigned long long) * VP_STAMP_N * n_wgs, 0, hipMemcpyDeviceToHost);
}
#endif

template <bool QOUT, bool HAS_ADD>
__global__ __launch_bounds__(512, 1) void value_proj_kernel(const ValueProjArgs g)
{
    extern __shared__ char vp_lds[];
    value_proj_body<QOUT, HAS_ADD>(g, vp_lds, vp_lds + VP_RAW0, reinterpret_cast<unsigned *>(vp_lds + VP_LDS));
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
    a.HW = HW; a.frames = frames; a.blocks = HW / VP_ROWS;
    a.x_bytes = (unsigned)((long)frames * 256 * HW * 4);
    // one workgroup per CU (its weights fill the register file); the pixel blocks of a frame are dealt round-robin
    const unsigned grid = (unsigned)(a.blocks < 256 ? a.blocks : 256);
#define VP_GO(Q_, A_, ATTR_, LDS_)                                                                                                 \
    do {                                                                                                                           \
        if (const int rc_attr = rac_set_dynamic_lds_once(ATTR_, reinterpret_cast<const void *>(value_proj_kernel<Q_, A_>), (int)(LDS_))) \
            return rc_attr;                                                                                                        \
        hipLaunchKernelGGL((value_proj_kernel<Q_, A_>), dim3(grid), dim3(512), LDS_, (hipStream_t)stream, a);                        \
    } while (0)
    if (q && add) VP_GO(true, true, RAC_ATTR_VALUE_PROJ_Q16, VP_LDS_Q16 + VP_STAMP_LDS);
    else if (q) VP_GO(true, false, RAC_ATTR_VALUE_PROJ_Q16_BIAS, VP_LDS_Q16 + VP_STAMP_LDS);
    else if (add) VP_GO(false, true, RAC_ATTR_VALUE_PROJ, VP_LDS + (VP_STAMP_LDS ? VP_LDS_Q16 - VP_LDS + VP_STAMP_LDS : 0));
    else VP_GO(false, false, RAC_ATTR_VALUE_PROJ_BIAS, VP_LDS + (VP_STAMP_LDS ? VP_LDS_Q16 - VP_LDS + VP_STAMP_LDS : 0));
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
