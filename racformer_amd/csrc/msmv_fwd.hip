// msmv_fwd.hip -- multi-scale multi-view sampling, forward, for gfx950 (MI355X).
//
// Replaces the reference's ms_deformable_im2col_gpu_kernel_{c45,c2345,c23456}
// (models/csrc/msmv_sampling/msmv_sampling_forward.cu:75-334) behind rac_msmv_fwd.
// Semantics (same as the reference kernel, :105-157 and :27-73): per point, view =
// round(loc_v*(N-1)); per level h_im = v*(H-1), w_im = u*(W-1) (align_corners=True), level
// skipped unless h_im>-1 && w_im>-1 && h_im<H && w_im<W; 4 bilinear taps, each bounds-checked,
// zero padding; levels accumulated in order c2..c5 with their scale weight.
//
// CDNA4 mapping (not the reference's thread-per-channel scheme):
//  * C=64 fast path: a 16-lane group owns one sampling point; each lane holds 4 adjacent
//    channels, so every tap is one 16-byte load per lane and one wave-instruction fetches four
//    256-byte pixel rows (1 KiB).  A wave64 walks the P points of one (slot, query) row four at
//    a time; all 4 levels x 4 taps = 16 independent loads are in flight per lane before the
//    first FMA.  No cross-lane reduction is needed: the reduction over levels/taps is in-register.
//  * the bilinear footprint of a point is computed ONCE (one thread per point, into an LDS tap table: four byte
//    offsets + four weights per level), not by each of the 16 lanes that gather it; the taps are buffer-descriptor
//    loads whose range check zero-fills taps outside the map (no branches), the accumulation is packed FMAs.
//  * block -> (slot, query block) mapping is XCD-aware: blocks b and b+8 share an XCD (and its
//    4 MiB L2), so XCD x walks slots x, x+8, ... and the 32 CUs of one XCD work on the same
//    slot's 23 MB pyramid at the same time, neighbouring queries (adjacent rays) together.
//  * RAC_OUT_BQGTPC writes the consumer layout directly: each 16-lane group stores one 256-byte
//    pixel row, 1 KiB contiguous per wave-instruction (the reference's P-minor layout needs
//    48-byte-strided 4-byte stores and a separate 88 MB permute afterwards).
#include "rac_common.h"

struct MsmvArgs {
    const void *feat[RAC_MAX_LEVELS];
    int H[RAC_MAX_LEVELS];
    int W[RAC_MAX_LEVELS];
    unsigned feat_bytes[RAC_MAX_LEVELS];   // size of one slot's N maps of each level (the buffer descriptors' ranges; C = 64 path)
    const float *loc;
    const float *w;
    float *out;
    int L, S, N, Q, P, C;
    int T, G;
    int blocks_per_slot;
};

#define MSMV_ROWS 8 /* (slot,query) rows per 256-thread workgroup: two per wave */
#define MSMV_TAP_OUTSIDE 0x80000000u   /* tap offset past the end of a level's buffer: the buffer load returns zeros */
typedef float msmv_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int msmv_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int msmv_u4 __attribute__((ext_vector_type(4)));

// Four channels of one tap through the level's buffer descriptor: its range check stands in for the four branches of the
// bilinear footprint (a tap outside the map carries the offset MSMV_TAP_OUTSIDE and reads as zero).
#ifndef MSMV_NT_LEVEL0
#define MSMV_NT_LEVEL0 1   /* measured on SURVEY 8d's uniform-stress set: 122 -> 112 us per launch (60 -> 65 % of the roofline) */
#endif
// AUX: cache policy of the load (0 default, 2 = nt: a line that is read once -- the finest level's taps on a scattered set --
// does not push the coarser levels' maps out of the XCD's L2)
template <typename FT, int AUX>
__device__ __forceinline__ rac_f4 msmv_tap(__amdgpu_buffer_rsrc_t rsrc, unsigned off);
template <>
__device__ __forceinline__ rac_f4 msmv_tap<float, 0>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    return __builtin_bit_cast(rac_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
}
template <>
__device__ __forceinline__ rac_f4 msmv_tap<float, 2>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    return __builtin_bit_cast(rac_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2));
}
template <>
__device__ __forceinline__ rac_f4 msmv_tap<unsigned short, 2>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    const msmv_u2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 2);    // 4 x bf16
    return (rac_f4){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                    __uint_as_float(r.y & 0xffff0000u)};
}
template <>
__device__ __forceinline__ rac_f4 msmv_tap<unsigned short, 0>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    const msmv_u2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);    // 4 x bf16
    return (rac_f4){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                    __uint_as_float(r.y & 0xffff0000u)};
}

template <typename FT, int L, bool OUT_CL>
__global__ __launch_bounds__(256, (L <= 4 ? 4 : 3)) void msmv_fwd_c64_kernel(const MsmvArgs a)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, sub = lane >> 4, c4 = lane & 15;
    const int P = a.P;

    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = xcd + 8 * (j / a.blocks_per_slot);
    if (s >= a.S)
        return;
    const int q0 = (j % a.blocks_per_slot) * MSMV_ROWS;
    const int nrows = min(MSMV_ROWS, a.Q - q0);

    // tap table [L][rows*P][8]: per point and level the 4 tap byte offsets into the level's buffer (MSMV_TAP_OUTSIDE =
    // outside the map) and the 4 bilinear weights with the level's scale weight folded in -- one thread per point builds it
    // from the op's loc / weight rows, instead of each of the 16 lanes that gather the point
    float *stab = smem;
    const int lstride = MSMV_ROWS * P * 8;
    const size_t row0 = (size_t)s * a.Q + q0;
    for (int i = tid; i < nrows * P; i += 256) {
        // (products and differences rounded one by one, as the reference kernel's statements read and as the oracle computes them:
        //  a contracted fma(v, H-1, -floor) moves a bilinear weight by an ulp of the pixel coordinate -- 1e-5 of a pixel at W = 176)
#pragma clang fp contract(off)
        const float *lp = a.loc + (row0 * P + i) * 3;
        const float *wp = a.w + (row0 * P + i) * L;
        const float lu = lp[0], lv = lp[1];
        int view = (int)roundf(lp[2] * (float)(a.N - 1));
        view = min(max(view, 0), a.N - 1);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int H = a.H[l], W = a.W[l];
            const float h_im = lv * (float)(H - 1);
            const float w_im = lu * (float)(W - 1);
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf;
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool t_ok = in && h_low >= 0, b_ok = in && h_high <= H - 1;
            const bool l_ok = w_low >= 0, r_ok = w_high <= W - 1;
            const unsigned pix_bytes = (unsigned)(64 * sizeof(FT));
            const unsigned mbase = (unsigned)view * (unsigned)(H * W) * pix_bytes;   // camera's map inside the slot's block of the level
            msmv_u4 off;
            off.x = t_ok && l_ok ? mbase + (unsigned)(h_low * W + w_low) * pix_bytes : MSMV_TAP_OUTSIDE;
            off.y = t_ok && r_ok ? mbase + (unsigned)(h_low * W + w_high) * pix_bytes : MSMV_TAP_OUTSIDE;
            off.z = b_ok && l_ok ? mbase + (unsigned)(h_high * W + w_low) * pix_bytes : MSMV_TAP_OUTSIDE;
            off.w = b_ok && r_ok ? mbase + (unsigned)(h_high * W + w_high) * pix_bytes : MSMV_TAP_OUTSIDE;
            const float wl = wp[l];
            float *e = stab + l * lstride + i * 8;
            *reinterpret_cast<msmv_u4 *>(e) = off;
            *reinterpret_cast<rac_f4 *>(e + 4) = (rac_f4){hh * hw * wl, hh * lw * wl, lh * hw * wl, lh * lw * wl};
        }
    }
    __syncthreads();
    // (the slot index through readfirstlane: the descriptor base then is scalar arithmetic -- left to itself hipcc computes it in
    //  vector registers and wraps every buffer load in a waterfall loop over a descriptor it can no longer prove uniform: +6 us)
    const int s_uni = __builtin_amdgcn_readfirstlane(s);
    __amdgpu_buffer_rsrc_t rsrc[L];
#pragma unroll
    for (int l = 0; l < L; ++l)
        // (one descriptor per level over THIS SLOT's N maps: offsets are relative to the slot, so only a slot's bytes -- not the whole
        //  level's, B * T * G slots -- have to stay below the 31-bit tap offsets)
        rsrc[l] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(a.feat[l]) + (size_t)s_uni * a.feat_bytes[l]), 0,
                                                    a.feat_bytes[l], 0x00020000);
    const unsigned lane_off = (unsigned)(c4 * 4 * sizeof(FT));
    // wave w gathers rows w, w+4 of the workgroup; per tap: one add for the lane's channel offset, one buffer load, two
    // packed FMAs; 16 taps (4 levels x 4) in flight per lane
    for (int row = wave; row < nrows; row += 4) {
        const int q = q0 + row;
        size_t out_row;  // element offset of this row's output block
        if (OUT_CL) {
            const int g = s % a.G, t = (s / a.G) % a.T, b = s / (a.G * a.T);
            out_row = ((((size_t)b * a.Q + q) * a.G + g) * a.T + t) * (size_t)P * 64;
        } else {
            out_row = ((size_t)s * a.Q + q) * 64 * (size_t)P;
        }
        for (int p0 = 0; p0 < P; p0 += 4) {
            const int p = p0 + sub;
            const bool act = p < P;
            const float *e = stab + (row * P + (act ? p : P - 1)) * 8;
            rac_f4 v[L][4], tw[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const msmv_u4 o = *reinterpret_cast<const msmv_u4 *>(e + l * lstride);
                tw[l] = *reinterpret_cast<const rac_f4 *>(e + l * lstride + 4);
                if (l == 0 && MSMV_NT_LEVEL0) {
                    v[l][0] = msmv_tap<FT, 2>(rsrc[l], o.x + lane_off);
                    v[l][1] = msmv_tap<FT, 2>(rsrc[l], o.y + lane_off);
                    v[l][2] = msmv_tap<FT, 2>(rsrc[l], o.z + lane_off);
                    v[l][3] = msmv_tap<FT, 2>(rsrc[l], o.w + lane_off);
                } else {
                    v[l][0] = msmv_tap<FT, 0>(rsrc[l], o.x + lane_off);
                    v[l][1] = msmv_tap<FT, 0>(rsrc[l], o.y + lane_off);
                    v[l][2] = msmv_tap<FT, 0>(rsrc[l], o.z + lane_off);
                    v[l][3] = msmv_tap<FT, 0>(rsrc[l], o.w + lane_off);
                }
            }
#if defined(RAC_DIAGNOSTIC_BUILD) && defined(RAC_GATHER_LOADS_FIRST)
            __builtin_amdgcn_sched_barrier(0);   // diagnostic (tools/race_victims.py): every tap load is issued before the first FMA
#endif
            rac_acc4 acc4 = rac_acc4_zero();
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float w4[4] = {tw[l].x, tw[l].y, tw[l].z, tw[l].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    rac_tap_fma(acc4, v[l][c].x, v[l][c].y, v[l][c].z, v[l][c].w, w4[c]);
                }
            }
            if (act) {
                rac_f4 r;
                rac_acc4_get(acc4, r.x, r.y, r.z, r.w);
                if (OUT_CL) {
                    *reinterpret_cast<rac_f4 *>(a.out + out_row + (size_t)p * 64 + c4 * 4) = r;
                } else {
                    float *o = a.out + out_row + (size_t)(c4 * 4) * P + p;
                    o[0] = r.x;
                    o[(size_t)P] = r.y;
                    o[(size_t)2 * P] = r.z;
                    o[(size_t)3 * P] = r.w;
                }
            }
        }
    }
}

// Any C / any L<=8: one thread per (row, point, channel), channel fastest (coalesced taps).
template <typename FT>
__device__ __forceinline__ float msmv_ld1(const FT *p);
template <>
__device__ __forceinline__ float msmv_ld1<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float msmv_ld1<unsigned short>(const unsigned short *p) { return rac_bf16_to_f32(*p); }

template <typename FT>
__global__ __launch_bounds__(256) void msmv_fwd_generic_kernel(const MsmvArgs a)
{
    const long total = (long)a.S * a.Q * a.P * a.C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % a.C);
        const long rp = idx / a.C;
        const int p = (int)(rp % a.P);
        const long r = rp / a.P;
        const int s = (int)(r / a.Q), q = (int)(r % a.Q);
        const float *lp = a.loc + rp * 3;
        const float *wp = a.w + rp * a.L;
        const float lu = lp[0], lv = lp[1];
        int view = (int)roundf(lp[2] * (float)(a.N - 1));
        view = min(max(view, 0), a.N - 1);
        float acc = 0.f;
        for (int l = 0; l < a.L; ++l) {
            const int H = a.H[l], W = a.W[l];
            const float h_im = lv * (float)(H - 1), w_im = lu * (float)(W - 1);
            if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W))
                continue;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const FT *base = (const FT *)a.feat[l] + ((size_t)s * a.N + view) * H * W * a.C + c;
            float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
            if (h_low >= 0 && w_low >= 0) v1 = msmv_ld1(base + ((size_t)h_low * W + w_low) * a.C);
            if (h_low >= 0 && w_high <= W - 1) v2 = msmv_ld1(base + ((size_t)h_low * W + w_high) * a.C);
            if (h_high <= H - 1 && w_low >= 0) v3 = msmv_ld1(base + ((size_t)h_high * W + w_low) * a.C);
            if (h_high <= H - 1 && w_high <= W - 1) v4 = msmv_ld1(base + ((size_t)h_high * W + w_high) * a.C);
            acc += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * wp[l];
        }
        size_t o;
        if (a.T > 0) {  // RAC_OUT_BQGTPC
            const int g = s % a.G, t = (s / a.G) % a.T, b = s / (a.G * a.T);
            o = (((((size_t)b * a.Q + q) * a.G + g) * a.T + t) * a.P + p) * a.C + c;
        } else {
            o = ((size_t)r * a.C + c) * a.P + p;
        }
        a.out[o] = acc;
    }
}

template <typename FT, bool OUT_CL>
static int launch_c64(const MsmvArgs &a, hipStream_t st)
{
    const int nb = 8 * ((a.S + 7) / 8) * a.blocks_per_slot;
    const size_t lds = (size_t)MSMV_ROWS * a.P * 8 * a.L * sizeof(float);
    switch (a.L) {
    case 2: hipLaunchKernelGGL((msmv_fwd_c64_kernel<FT, 2, OUT_CL>), dim3(nb), dim3(256), lds, st, a); break;
    case 4: hipLaunchKernelGGL((msmv_fwd_c64_kernel<FT, 4, OUT_CL>), dim3(nb), dim3(256), lds, st, a); break;
    case 5: hipLaunchKernelGGL((msmv_fwd_c64_kernel<FT, 5, OUT_CL>), dim3(nb), dim3(256), lds, st, a); break;
    default: return 1;
    }
    return 0;
}

extern "C" int rac_msmv_fwd(const void *const *feats, const int32_t *hw, int L, const float *loc,
                            const float *w, float *out, int S, int N, int Q, int P, int C, int dtype,
                            int out_layout, int T, int G, void *stream)
{
    RAC_CHECK_ARG(L >= 1 && L <= RAC_MAX_LEVELS, "rac_msmv_fwd: L=%d out of [1,%d]", L, RAC_MAX_LEVELS);
    RAC_CHECK_ARG(S >= 0 && Q >= 0 && N >= 1 && C >= 1, "rac_msmv_fwd: bad sizes S=%d N=%d Q=%d C=%d", S, N, Q, C);
    RAC_CHECK_ARG(P >= 0 && P <= RAC_MAX_POINTS, "rac_msmv_fwd: num_point exceed limits (P=%d > %d)", P, RAC_MAX_POINTS);
    if (S == 0 || Q == 0 || P == 0)
        return 0;  // empty output: nothing to launch (torch hands out null data pointers for empty tensors)
    RAC_CHECK_ARG(feats && hw && loc && w && out, "rac_msmv_fwd: null pointer");
    RAC_CHECK_ARG(dtype == RAC_F32 || dtype == RAC_BF16, "rac_msmv_fwd: dtype %d", dtype);
    RAC_CHECK_ARG(out_layout == RAC_OUT_SQCP || out_layout == RAC_OUT_BQGTPC, "rac_msmv_fwd: layout %d", out_layout);
    if (out_layout == RAC_OUT_BQGTPC)
        RAC_CHECK_ARG(T >= 1 && G >= 1 && S % (T * G) == 0, "rac_msmv_fwd: S=%d not a multiple of T*G=%d*%d", S, T, G);
    MsmvArgs a;
    for (int l = 0; l < L; ++l) {
        RAC_CHECK_ARG(feats[l] != nullptr, "rac_msmv_fwd: feats[%d] is null", l);
        RAC_CHECK_ARG(hw[2 * l] >= 1 && hw[2 * l + 1] >= 1, "rac_msmv_fwd: level %d has empty map", l);
        a.feat[l] = feats[l];
        a.H[l] = hw[2 * l];
        a.W[l] = hw[2 * l + 1];
        const size_t bytes = (size_t)N * a.H[l] * a.W[l] * C * (dtype == RAC_F32 ? 4 : 2);        // one slot's maps
        a.feat_bytes[l] = bytes < (size_t)MSMV_TAP_OUTSIDE ? (unsigned)bytes : 0u;    // 0: too large for the 31-bit tap offsets
    }
    for (int l = L; l < RAC_MAX_LEVELS; ++l) {
        a.feat[l] = nullptr;
        a.H[l] = a.W[l] = 1;
        a.feat_bytes[l] = 0;
    }
    bool small_maps = true;
    for (int l = 0; l < L; ++l)
        small_maps = small_maps && a.feat_bytes[l] != 0;
    a.loc = loc; a.w = w; a.out = out;
    a.L = L; a.S = S; a.N = N; a.Q = Q; a.P = P; a.C = C;
    a.T = out_layout == RAC_OUT_BQGTPC ? T : 0;
    a.G = out_layout == RAC_OUT_BQGTPC ? G : 1;
    a.blocks_per_slot = (Q + MSMV_ROWS - 1) / MSMV_ROWS;
    hipStream_t st = (hipStream_t)stream;
    int fell_through = 1;
    if (C == 64 && (L == 2 || L == 4 || L == 5) && small_maps && (size_t)MSMV_ROWS * P * 8 * L * sizeof(float) <= 64 * 1024) {
        if (dtype == RAC_F32)
            fell_through = out_layout == RAC_OUT_BQGTPC ? launch_c64<float, true>(a, st) : launch_c64<float, false>(a, st);
        else
            fell_through = out_layout == RAC_OUT_BQGTPC ? launch_c64<unsigned short, true>(a, st)
                                                        : launch_c64<unsigned short, false>(a, st);
    }
    if (fell_through) {
        const long total = (long)S * Q * P * C;
        const int nb = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        if (dtype == RAC_F32)
            hipLaunchKernelGGL(msmv_fwd_generic_kernel<float>, dim3(nb), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL(msmv_fwd_generic_kernel<unsigned short>, dim3(nb), dim3(256), 0, st, a);
    }
    return rac_launch_status("rac_msmv_fwd");
}
