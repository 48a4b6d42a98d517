// sampling_fused.hip -- adaptive 4D sampling as ONE kernel, for gfx950 (MI355X).
//
// Fuses, per decoder layer, what the reference runs as ~60 elementwise torch kernels, a
// broadcast-expanded [B,T,N,Q,GP,4,4] matmul, two permute copies and the msmv gather:
//   RaCFormerSampling.inner_forward  models/racformer_transformer.py:361-408  (keypoints)
//   sampling_4d                      models/sparsebev_sampling.py:45-131      (projection, validity,
//                                    first-valid-view selection, slot regroup, output regroup)
//   msmv op                          models/csrc/msmv_sampling/msmv_sampling_forward.cu:75-164
// Inputs are the three Linear outputs of the sampling module (offsets, ray-depth logits, scale
// logits), the query boxes, time_diff and lidar2img; output is [B,Q,G,T*P,C] (what sampling_4d
// returns).  Nothing else touches HBM: locations and softmaxed scale weights live in LDS only
// (optionally also written out for parity debugging, the counterpart of the reference's DUMP
// hooks, sparsebev_sampling.py:83-87).
//
// Workgroup = 4 waves = S4D_ROWS consecutive queries of one slot (b,t,g).  Phase 1: the first S4D_ROWS*P threads
// compute one keypoint each (box decode, offset, yaw rotation, velocity warp, polar jitter,
// projection into the N cameras of frame t, first valid view, softmax over levels) and its tap table (per level: four
// byte offsets + four weights) into LDS.
// Phase 2: 16-lane group per point, 16-byte buffer loads (the descriptor's range check zero-fills taps outside the map),
// 16 taps in flight, packed FMAs, XCD-aware slot mapping, 1 KiB coalesced stores.
//
// Index conventions reproduced from the reference: point p of a group = (num_point, depth) with
// depth fastest (:394); slot s = (b*T+t)*G+g for points/features/outputs, but the scale weights of
// slot s are read from the (b,g',t') flattening at s' = t*G+g, g' = s'/T, t' = s'%T
// (sparsebev_sampling.py:113-120, quirk Q1).
#include "rac_common.h"

#ifndef S4D_ROWS
#define S4D_ROWS 8 /* queries per workgroup: one prologue pass (96 keypoints at P = 12) serves 2 gather rounds per wave; measured 4 / 8 / 12 / 16 / 32: 92.7 / 88.9 / 90.3 / 94 / 107 us */
#endif
#ifndef S4D_LB
#define S4D_LB 4 /* levels per load batch (see the gather loop) */
#endif
#ifndef S4D_WPS
#define S4D_WPS 4 /* waves per SIMD the register allocator must allow */
#endif
#include "s4d_device.h"

// COMPACT: a row's points with no tap inside any map (no camera sees them: about half of the points of a 3-camera rig) are
// set aside -- the gather walks only the others, four per wave-step, and the rest of the row is zero-filled by plain stores.
template <typename FT, int L, bool COMPACT>
__global__ __launch_bounds__(256, (L <= 4 ? S4D_WPS : 3)) void sampling4d_c64_kernel(const S4dArgs a)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, sub = lane >> 4, c4 = lane & 15;
    const int P = a.P;

    const int S = a.B * a.T * a.G;
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = xcd + 8 * (j / a.blocks_per_slot);
    if (s >= S)
        return;
    const int q0 = (j % a.blocks_per_slot) * a.rows;
    const int nrows = min(a.rows, a.Q - q0);
    const int g = s % a.G, t = (s / a.G) % a.T, b = s / (a.G * a.T);

    // tap table [L][rows*P][8]: per keypoint and level the 4 tap byte offsets into the level's buffer (S4D_TAP_OUTSIDE =
    // outside the map) and the 4 bilinear weights with the level weight folded in -- computed once per keypoint by the
    // prologue instead of by each of the 16 lanes that gather the point
    float *stab = smem;
    const int lstride = a.rows * P * 8;
    float *sl2i = stab + L * lstride;       // [N][16]
    unsigned char *sval = reinterpret_cast<unsigned char *>(sl2i + a.N * 16);   // [rows*P] 1: some tap of the point is inside a map
    unsigned char *sperm = sval + a.rows * P;                                    // [rows][P] points with taps first, then the others
    for (int i = tid; i < a.N * 16; i += 256)
        sl2i[i] = a.l2i[((size_t)b * a.T + t) * a.N * 16 + i];
    __syncthreads();
    for (int i = tid; i < nrows * P; i += 256) {
        const int r = i / P, p = i - r * P;
        float loc3[3], wl[L];
        s4d_keypoint<L>(a, sl2i, b, t, g, q0 + r, p, loc3, wl);
        const float lu = loc3[0], lv = loc3[1];
        const int view = (int)loc3[2] & 255;
        bool any = false;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            any = s4d_taps_of_level<FT>(a.H[l], a.W[l], lu, lv, view, wl[l], 0u, stab + l * lstride + i * 8) || any;
        }
        if (COMPACT)
            sval[i] = any ? 1 : 0;
        if (a.loc_out) {
            float *lo = a.loc_out + (((size_t)s * a.Q + q0 + r) * P + p) * 3;
            lo[0] = lu;
            lo[1] = lv;
            // loc_out reports the kernel's own first-valid-view choice (with view_in: beside the imposed one it sampled in)
            lo[2] = (float)((int)loc3[2] >> 8) / (float)max(a.N - 1, 1);
            float *wo = a.w_out + (((size_t)s * a.Q + q0 + r) * P + p) * L;
#pragma unroll
            for (int l = 0; l < L; ++l)
                wo[l] = wl[l];
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
    // wave w gathers rows w, w+4, ... of the workgroup's S4D_ROWS queries: the keypoint prologue above (one pass, its
    // latency independent of the number of keypoints up to 256) is paid once per S4D_ROWS / 4 gather rounds.  Per tap:
    // one add for the lane's channel offset, one buffer load, two packed FMAs.
    for (int row = wave; row < nrows; row += 4) {
        const int q = q0 + row;
        const size_t out_row = ((((size_t)b * a.Q + q) * a.G + g) * a.T + t) * (size_t)P * 64;
        int nlive = P;
        if (COMPACT) {
            // rank the row's points inside the wave: lanes < P hold one point each (P <= 64)
            const bool mine = lane < P && sval[row * P + lane] != 0;
            const unsigned long long mask = __ballot(mine);
            nlive = __popcll(mask);
            if (lane < P) {
                const int before = __popcll(mask & ((1ull << lane) - 1ull));
                sperm[row * P + (mine ? before : nlive + (lane - before))] = (unsigned char)lane;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (same wave reads it back)
            // points without any tap: zeros, four rows of 256 B per wave-step
            for (int z0 = nlive; z0 < P; z0 += 4) {
                if (z0 + sub < P) {
                    const int pz = sperm[row * P + z0 + sub];
                    *reinterpret_cast<rac_f4 *>(a.out + out_row + (size_t)pz * 64 + c4 * 4) = (rac_f4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        for (int p0 = 0; p0 < nlive; p0 += 4) {
            const bool act = p0 + sub < nlive;
            const int p = COMPACT ? (int)sperm[row * P + (act ? p0 + sub : nlive - 1)] : (act ? p0 + sub : P - 1);
            const float *e = stab + (row * P + p) * 8;
            rac_f4 v[L][4], tw[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const s4d_u4 o = *reinterpret_cast<const s4d_u4 *>(e + l * lstride);
                tw[l] = *reinterpret_cast<const rac_f4 *>(e + l * lstride + 4);
                v[l][0] = s4d_tap<FT>(rsrc[l], o.x + lane_off);
                v[l][1] = s4d_tap<FT>(rsrc[l], o.y + lane_off);
                v[l][2] = s4d_tap<FT>(rsrc[l], o.z + lane_off);
                v[l][3] = s4d_tap<FT>(rsrc[l], o.w + lane_off);
            }
            rac_acc4 acc4 = rac_acc4_zero();
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float w4[4] = {tw[l].x, tw[l].y, tw[l].z, tw[l].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    rac_tap_fma(acc4, v[l][c].x, v[l][c].y, v[l][c].z, v[l][c].w, w4[c]);
                }
            }
            if (act)
            {
                rac_f4 r;
                rac_acc4_get(acc4, r.x, r.y, r.z, r.w);
                *reinterpret_cast<rac_f4 *>(a.out + out_row + (size_t)p * 64 + c4 * 4) = r;
            }
        }
    }
}

extern "C" int rac_sampling4d_fwd(const void *const *feats, const int32_t *hw, int L, const float *query_bbox,
                                  const float *box_table,
                                  const float *offsets, const float *ray_logits, const float *scale_logits,
                                  const float *time_diff, const float *lidar2img, float *out, float *loc_out,
                                  float *w_out, const unsigned char *view_in, int ld_off, int ld_ray, int ld_scale, int B, int T, int N, int G, int Q, int NP,
                                  int D, int C,
                                  const float *pc_range, const float *depth_base, float d_region, float image_h,
                                  float image_w, float eps, int dtype, int compact, void *stream)
{
    RAC_CHECK_ARG(compact >= -1 && compact <= 1, "rac_sampling4d_fwd: compact=%d (-1 automatic, 0 plain, 1 compact)", compact);
    RAC_CHECK_ARG(L == 2 || L == 4 || L == 5, "rac_sampling4d_fwd: L=%d (supported: 2, 4, 5)", L);
    RAC_CHECK_ARG(C == 64, "rac_sampling4d_fwd: C=%d (the fused kernel is built for 64 channels per group)", C);
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && T >= 1 && N >= 1 && N <= S4D_MAX_CAMS && G >= 1 && NP >= 1 && D >= 1 &&
                      D <= S4D_MAX_DEPTH,
                  "rac_sampling4d_fwd: bad sizes B=%d T=%d N=%d G=%d Q=%d NP=%d D=%d", B, T, N, G, Q, NP, D);
    const int P = NP * D;
    RAC_CHECK_ARG(P <= RAC_MAX_POINTS, "rac_sampling4d_fwd: num_point exceed limits (P=%d > %d)", P, RAC_MAX_POINTS);
    RAC_CHECK_ARG(dtype == RAC_F32 || dtype == RAC_BF16, "rac_sampling4d_fwd: dtype %d", dtype);
    if (B == 0 || Q == 0)
        return 0;
    RAC_CHECK_ARG(box_table != nullptr, "rac_sampling4d_fwd: box_table is null (run rac_box_prep_fwd first)");
    RAC_CHECK_ARG(feats && hw && query_bbox && offsets && ray_logits && scale_logits && time_diff && lidar2img &&
                      out && pc_range && depth_base,
                  "rac_sampling4d_fwd: null pointer");
    RAC_CHECK_ARG((loc_out == nullptr) == (w_out == nullptr), "rac_sampling4d_fwd: loc_out and w_out go together");
    RAC_CHECK_ARG(ld_off >= G * NP * D * 3 && ld_ray >= D && ld_scale >= G * T * NP * D * L, "rac_sampling4d_fwd: row strides too small");
    S4dArgs a;
    for (int l = 0; l < RAC_MAX_LEVELS; ++l) {
        a.feat[l] = nullptr;
        a.H[l] = a.W[l] = 1;
        a.feat_bytes[l] = 0;
    }
    for (int l = 0; l < L; ++l) {
        RAC_CHECK_ARG(feats[l] != nullptr && hw[2 * l] >= 1 && hw[2 * l + 1] >= 1, "rac_sampling4d_fwd: level %d", l);
        a.feat[l] = feats[l];
        a.H[l] = hw[2 * l];
        a.W[l] = hw[2 * l + 1];
        const size_t bytes = (size_t)N * a.H[l] * a.W[l] * 64 * (dtype == RAC_F32 ? 4 : 2);      // one slot's maps
        RAC_CHECK_ARG(bytes < (size_t)S4D_TAP_OUTSIDE, "rac_sampling4d_fwd: a slot of level %d holds %zu bytes (the tap offsets are 31-bit)", l, bytes);
        a.feat_bytes[l] = (unsigned)bytes;
    }
    a.qbox = query_bbox; a.box = box_table; a.off = offsets; a.ray = ray_logits; a.scale = scale_logits;
    a.time_diff = time_diff; a.l2i = lidar2img; a.out = out; a.loc_out = loc_out; a.w_out = w_out; a.view_in = view_in;
    for (int i = 0; i < S4D_MAX_DEPTH; ++i)
        a.depth_base[i] = i < D ? depth_base[i] : 0.f;
    for (int i = 0; i < 6; ++i)
        a.pc[i] = pc_range[i];
    a.d_region = d_region; a.image_h = image_h; a.image_w = image_w; a.eps = eps;
    a.L = L; a.B = B; a.T = T; a.N = N; a.G = G; a.Q = Q; a.NP = NP; a.D = D; a.P = P;
    a.ld_off = ld_off; a.ld_ray = ld_ray; a.ld_scale = ld_scale;
    // queries per workgroup: S4D_ROWS, fewer where the tap table of that many rows (rows * P * L * 32 bytes) would not fit 64 KB
    // (P = 64 with four levels: 4 rows; P = 128 with five levels: 2)
    a.rows = S4D_ROWS;
    auto lds_bytes = [&](int rows) { return ((size_t)rows * P * 8 * L + (size_t)N * 16) * sizeof(float) + 2 * (size_t)rows * P; };
    while (a.rows > 1 && lds_bytes(a.rows) > 64 * 1024)
        a.rows >>= 1;
    a.blocks_per_slot = (Q + a.rows - 1) / a.rows;
    const int S = B * T * G;
    const int nb = 8 * ((S + 7) / 8) * a.blocks_per_slot;
    const size_t lds = lds_bytes(a.rows);
    RAC_CHECK_ARG(lds <= 64 * 1024, "rac_sampling4d_fwd: P=%d x L=%d too large for the LDS tap table", P, L);
    hipStream_t st = (hipStream_t)stream;
    // Rigs whose cameras do not cover the full circle leave many points without any tap; the COMPACT variant walks only the others
    // (3-cam rig: 76.8 -> 67.9 us) and costs 2.5 us where every point is live.  The caller decides from the rig's measured
    // coverage (RaCFormerTransformerDecoder.stage_metas: share of a ring of probe points some camera sees); -1 = no measurement
    // at hand: by the number of cameras.
#ifndef S4D_COMPACT_MAX_CAMS
#define S4D_COMPACT_MAX_CAMS 3
#endif
    const bool use_compact = (compact < 0 ? N <= S4D_COMPACT_MAX_CAMS : compact == 1) && P <= 64;
#define S4D_LAUNCH(FT, LL)                                                                                           \
    do {                                                                                                             \
        if (use_compact) hipLaunchKernelGGL((sampling4d_c64_kernel<FT, LL, true>), dim3(nb), dim3(256), lds, st, a);     \
        else hipLaunchKernelGGL((sampling4d_c64_kernel<FT, LL, false>), dim3(nb), dim3(256), lds, st, a);            \
    } while (0)
    if (dtype == RAC_F32) {
        if (L == 2) S4D_LAUNCH(float, 2); else if (L == 4) S4D_LAUNCH(float, 4); else S4D_LAUNCH(float, 5);
    } else {
        if (L == 2) S4D_LAUNCH(unsigned short, 2); else if (L == 4) S4D_LAUNCH(unsigned short, 4); else S4D_LAUNCH(unsigned short, 5);
    }
#undef S4D_LAUNCH
    return rac_launch_status("rac_sampling4d_fwd");
}
