// bev_fused.hip -- BEV deformable cross-attention of one decoder layer as ONE kernel (gfx950).
//
// Fuses BEVSampling.inner_forward's keypoint chain (models/racformer_transformer.py:490-529),
// the per-frame multi-scale deformable attention (models/bev_self_attention.py:176-204, one level,
// heads x 20 points, Deformable-DETR bilinear semantics) and the learned frame fusion
// (softmax over the T BEV maps, :207-213) -- in the reference ~50 elementwise kernels, the MSDA op
// and a [B*T,Q,256] -> permute -> softmax -> sum chain per BEV stream per layer.
// Inputs: the three Linear outputs of the module (offsets, ray-depth logits, point-weight
// logits), the frame logits (bev_queue_weight), query boxes, time_diff and the hoisted value
// stream [B*T, H*W, heads, 64]; output [B,Q,heads*64] (before output_proj).
//
// Workgroup = 4 items (b,q,head) (one query's 4 heads) x 4 point subsets: a 16-lane group owns
// (item, points ts, ts+4, ... of every frame) with 4 channels per lane (16-byte loads).  Phase A: the T-invariant
// pieces once per item -- base points from the box table (rac_box_prep_fwd), ray-depth offsets, the
// two softmaxes -- into LDS.  Phase B: all threads warp the base points to every frame (T*P
// keypoints per item).  Phase C: per frame 20 points unrolled by 4 (16 taps in flight), scaled by
// the frame weight.  Phase D: fixed-order LDS sum of the four point subsets (deterministic).
// For B>1 the reference pairs value frame i=b*T+t with the locations of (t'=i/B, b'=i%B)
// (bev_self_attention.py:185-188 vs :162,173, quirk Q2); reproduced as written.
#include "rac_common.h"

#define BEV_MAX_DEPTH 16
#define BEV_TWO_PI 6.283185307179586f

#define BEV_MAX_STREAMS 2
// one BEV stream (radar / LSS): its value maps and the Linear outputs of its sampling module; blockIdx.y selects it, so
// the streams of a decoder layer -- same queries, same boxes -- share ONE launch: the second stream's workgroups start
// as the first one's drain, its keypoint prologue runs under the first stream's gathers.
struct BevStream {
    const void *value;
    const float *off;        // [B,Q,heads*P*2]
    const float *ray;        // [B,Q,D]
    const float *scale;      // [B,Q,heads*P] logits
    const float *queue;      // [B,Q,T] logits
    float *out;              // [B,Q,heads*64]
    float *loc_out;          // optional [B,Q,heads,T,P,2]
    const float *vscale;     // int16 block storage only: [B*T, H*W, heads] scale of each (pixel, head) block (quant.hip)
};
struct BevArgs {
    BevStream s[BEV_MAX_STREAMS];
    const float *qbox;       // [B,Q,10]
    const float *box;        // [B,Q,8] from rac_box_prep_fwd
    const float *time_diff;  // [B,T]
    float depth_base[BEV_MAX_DEPTH];
    float pc[6];
    float d_region;
    int B, T, Q, heads, NP, D, P, H, W;
    int ld_off, ld_ray, ld_scale, ld_queue;  // row strides (floats): slices of one fused GEMM output
    int blocks_per_b;
    unsigned value_bytes;    // size of one stream's value buffer (the buffer descriptor's range)
    int list_len;            // entries of a 16-lane group's tap list
    int list_holes;          // 1: some list slots have no keypoint (filled with zero-weight outside taps)
    int xcd_remap;           // 1: blocks that share an XCD (blockIdx & 7) take a contiguous range of items (speed only)
};

// polar jitter of a keypoint (racformer_transformer.py:512-522 through models/bbox/utils.py:84-106): (ex, ey) metres from the
// map centre -> (dist, theta), dist += doff, back to the normalised map.  (Scaling the unit vector (ex, ey) / r directly
// would skip atan2f / fmodf / cosf / sinf; measured: under 1 us of 81 per launch -- not worth leaving the reference's chain.)
__device__ __forceinline__ void bev_polar_jitter(float ex, float ey, float doff, float *loc2)
{
    const float dist = sqrtf(ex * ex + ey * ey) / 65.0f + doff;
    const float th = fmodf(atan2f(ey, ex) + BEV_TWO_PI, BEV_TWO_PI) / BEV_TWO_PI;
    const float ang = th * BEV_TWO_PI, rad = dist * 65.0f;
    loc2[0] = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
    loc2[1] = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
}

__device__ __forceinline__ void bev_keypoint(const BevArgs &a, const BevStream &s, int bq, int tq, int q, int h, int p, float *loc2)
{
    const float *qb = a.qbox + ((size_t)bq * a.Q + q) * 10;
    const float sx = a.pc[3] - a.pc[0], sy = a.pc[4] - a.pc[1];
    const float ang0 = qb[0] * BEV_TWO_PI, rad0 = qb[1] * 65.0f;
    const float xn0 = fminf(fmaxf((51.2f + rad0 * cosf(ang0)) / 102.4f, 0.f), 1.f);
    const float yn0 = fminf(fmaxf((51.2f + rad0 * sinf(ang0)) / 102.4f, 0.f), 1.f);
    const float cx = xn0 * sx + a.pc[0], cy = yn0 * sy + a.pc[1];
    const float yaw = atan2f(qb[6], qb[7]);
    const float cs = cosf(yaw), sn = sinf(yaw);
    const float *o = s.off + ((size_t)bq * a.Q + q) * a.ld_off + ((size_t)h * a.P + p) * 2;
    const float dx = expf(qb[3]) * o[0], dy = expf(qb[4]) * o[1];
    float px = cx + (dx * cs - dy * sn);
    float py = cy + (dx * sn + dy * cs);
    const float td = a.time_diff[bq * a.T + tq];
    px -= qb[8] * td;
    py -= qb[9] * td;
    const float nx = (px - a.pc[0]) / sx, ny = (py - a.pc[1]) / sy;
    const float ex = nx * 102.4f - 51.2f, ey = ny * 102.4f - 51.2f;
    const int dd = p % a.D;
    const float sg = 1.f / (1.f + expf(-s.ray[((size_t)bq * a.Q + q) * a.ld_ray + dd]));
    bev_polar_jitter(ex, ey, a.depth_base[dd] + (sg * 2.f - 1.f) * a.d_region / (float)a.D / 2.f, loc2);
}

#ifndef BEV_U
#define BEV_U 4    /* keypoints per gather batch (4 taps each in flight) */
#endif
#ifndef BEV_OCC
#define BEV_OCC 4  /* workgroups per CU the register budget is cut for */
#endif
#define BEV_GI 4   /* items (b,q,head) per workgroup */
#define BEV_TS 4   /* point subsets per item: 16-lane group (k, ts) handles points ts, ts+4, ... of every frame */
#define BEV_TAP_OUTSIDE 0x80000000u   /* tap offset past the end of the value buffer: the buffer load returns zeros */

// Diagnostic build only (tools/bev_phase_split.py; -DBEV_STAMPS): s_memtime at the phase boundaries of every workgroup (wave 0),
// kept in a device array of the code object and read back through rac_dbg_bev_stamps.  Not part of the product library or its ABI.
#ifdef BEV_STAMPS
#define BEV_STAMP_WGS 4096
__device__ unsigned long long bev_stamp_buf[BEV_STAMP_WGS * 8];
#define BEV_STAMP(i)                                                                                    \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < BEV_STAMP_WGS)                    \
            bev_stamp_buf[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BEV_STAMP(i)
#endif

typedef float bev_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int bev_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int bev_u4 __attribute__((ext_vector_type(4)));

// Four channels of one tap through a buffer descriptor: the range check of the descriptor stands in for the four
// branches of the bilinear footprint (a tap outside the map carries the offset BEV_TAP_OUTSIDE and reads as zero).
template <typename FT>
__device__ __forceinline__ rac_f4 bev_tap(__amdgpu_buffer_rsrc_t rsrc, unsigned off);
template <>
__device__ __forceinline__ rac_f4 bev_tap<float>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    return __builtin_bit_cast(rac_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
}
template <>
__device__ __forceinline__ rac_f4 bev_tap<unsigned short>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    const bev_u2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);    // 4 x bf16
    return (rac_f4){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                    __uint_as_float(r.y & 0xffff0000u)};
}

// per-(t,p) half of the keypoint chain for B==1: warp the T-invariant base point, polar jitter.
__device__ __forceinline__ void bev_warp(const BevArgs &a, float px, float py, float vx, float vy, float td,
                                         float doff, float *loc2)
{
    const float sx = a.pc[3] - a.pc[0], sy = a.pc[4] - a.pc[1];
    px -= vx * td;
    py -= vy * td;
    const float nx = (px - a.pc[0]) / sx, ny = (py - a.pc[1]) / sy;
    const float ex = nx * 102.4f - 51.2f, ey = ny * 102.4f - 51.2f;
    bev_polar_jitter(ex, ey, doff, loc2);
}

__device__ __forceinline__ float bev_wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float bev_wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// GL = lanes of a group (one group gathers one tap row per load): 16 lanes x 4 channels for the 4-byte and bf16 streams; 8 lanes x 8
// channels (16 bytes of int16 per lane) for the int16 block storage -- the texture path's cost is per wave-instruction (address work for
// 64 lanes) at least as much as per byte: with 16 lanes per tap the int16 stream halved the bytes at the same number of tap
// instructions and bought 9 % (80 -> 73 us); with 8 lanes per tap a wave-instruction covers 8 tap rows instead of 4.
template <typename FT, int GL>
__global__ __launch_bounds__(256, BEV_OCC) void bev_sampling_d64_kernel(const BevArgs a)
{
    constexpr int TSN = 64 / GL;      // groups per item (= per wave): keypoint subsets
    static_assert((GL == 16 && sizeof(FT) >= 2) || (GL == 8 && sizeof(FT) == 2), "16 lanes x 4 channels, or 8 lanes x 8 two-byte channels");
    extern __shared__ float smem[];
    BEV_STAMP(0);
    const BevStream &s = a.s[blockIdx.y];
    const int tid = threadIdx.x;
    const int c4 = tid & (GL - 1), grp = tid / GL;
    const int k = grp / TSN, ts = grp % TSN;    // item within the workgroup (= wave), keypoint subset
    const int T = a.T, P = a.P, TP = a.T * a.P, D = a.D;
    const int npp = (P + BEV_TS - 1) / BEV_TS;  // GL = 16: points of a subset per frame (the last subsets may hold one less)
    const int Lg = a.list_len;                  // entries of a group's tap list: T * npp rounded up to BEV_U

    const int per_b = a.Q * a.heads;
    int bid = blockIdx.x;
    if (a.xcd_remap) {
        // blocks id, id+8, id+16, ... run on one XCD (one 4 MiB L2): give them neighbouring queries -- neighbouring
        // rays of the polar query grid sample neighbouring BEV pixels.  Bijective for any grid size.
        const int nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7, x = bid & 7;
        bid = (x < rr ? x * (qq + 1) : rr * (qq + 1) + (x - rr) * qq) + (bid >> 3);
    }
    const int b = bid / a.blocks_per_b;
    const int i0 = (bid % a.blocks_per_b) * BEV_GI;
    const int nitems = min(BEV_GI, per_b - i0);

    // tap lists, one per 16-lane group and in the order the group walks them (frame-major): [GI][TS][Lg][8] =
    // 4 tap byte offsets into the value buffer (BEV_TAP_OUTSIDE = outside the map) + 4 tap weights
    float *stab = smem;
    float *spart = stab + BEV_GI * TSN * Lg * 8;     // [GI][TSN][64] partial sums
    const int Tw = a.B > 1 ? T : 1;            // B > 1: the point weights depend on the frame (paired batch, quirk Q2)
    float *sattn = spart + BEV_GI * TSN * 64;        // [GI][Tw][P]
    float *sq = sattn + BEV_GI * Tw * P;       // [GI][T]
    float *sbase = sq + BEV_GI * T;            // [GI][P][2]  T-invariant base points (B==1)
    float *sdoff = sbase + BEV_GI * P * 2;     // [GI][D]
    float *svel = sdoff + BEV_GI * BEV_MAX_DEPTH;    // [GI][2] query velocity, then [T] time_diff of batch b (B==1)
    float *std_ = svel + BEV_GI * 2;

    // list slots no keypoint fills (P not a multiple of 4, T * npp not a multiple of BEV_U): outside, weight 0
    if (a.list_holes) {
        for (int i = tid; i < BEV_GI * TSN * Lg * 2; i += 256) {
            const unsigned fill = (i & 1) ? 0u : BEV_TAP_OUTSIDE;
            reinterpret_cast<bev_u4 *>(stab)[i] = (bev_u4){fill, fill, fill, fill};
        }
    }

    // phase A: T-invariant pieces.  threads [0, GI*P): base points; [128,128+GI*D): depth offsets; wave k: the two
    // softmaxes of item k (point weights, frame weights) across its lanes.  Every role issues its global loads first --
    // one round trip for the whole phase instead of one per role (and per softmax term).
    const bool r_base = tid < nitems * P && a.B == 1;
    const bool r_doff = tid >= 128 && tid < 128 + nitems * D && a.B == 1;
    const int wk = tid >> 6, ln = tid & 63;
    const bool r_soft = wk < nitems;
    float g_bt[8], g_o0 = 0.f, g_o1 = 0.f, g_ray = 0.f, g_q = -INFINITY, g_lg = -INFINITY;
    if (r_base) {
        const int kk = tid / P, p = tid - kk * P;
        const int it = i0 + kk, q = it / a.heads, h = it % a.heads;
        const float *bt = a.box + ((size_t)b * a.Q + q) * 8;
        const float *o = s.off + ((size_t)b * a.Q + q) * a.ld_off + ((size_t)h * P + p) * 2;
        const rac_f4 b0 = rac_ld4(bt), b1 = rac_ld4(bt + 4);
        g_bt[0] = b0.x; g_bt[1] = b0.y; g_bt[3] = b0.w; g_bt[4] = b1.x; g_bt[6] = b1.z; g_bt[7] = b1.w;
        g_o0 = o[0]; g_o1 = o[1];
    }
    if (r_doff) {
        const int kk = (tid - 128) / D, dd = (tid - 128) - kk * D;
        const int q = (i0 + kk) / a.heads;
        g_ray = s.ray[((size_t)b * a.Q + q) * a.ld_ray + dd];
    }
    if (r_soft) {
        const int it = i0 + wk, q = it / a.heads, h = it % a.heads;
        if (ln < T)
            g_q = s.queue[((size_t)b * a.Q + q) * a.ld_queue + ln];
        if (a.B == 1 && ln < P)
            g_lg = s.scale[((size_t)b * a.Q + q) * a.ld_scale + (size_t)h * P + ln];
    }
    // velocity of each item's query and the frame times (phase B reads them per keypoint): threads 224.. / 192..
    if (a.B == 1 && tid >= 224 && tid < 224 + nitems * 2) {
        const int kk = (tid - 224) >> 1;
        svel[tid - 224] = a.qbox[((size_t)b * a.Q + (i0 + kk) / a.heads) * 10 + 8 + ((tid - 224) & 1)];
    }
    if (a.B == 1 && tid >= 192 && tid < 192 + min(T, 32))
        for (int t = tid - 192; t < T; t += 32)
            std_[t] = a.time_diff[b * T + t];
    if (r_base) {
        const float dx = g_bt[3] * g_o0, dy = g_bt[4] * g_o1;
        sbase[tid * 2] = g_bt[0] + (dx * g_bt[6] - dy * g_bt[7]);
        sbase[tid * 2 + 1] = g_bt[1] + (dx * g_bt[7] + dy * g_bt[6]);
    }
    if (r_doff) {
        const int kk = (tid - 128) / D, dd = (tid - 128) - kk * D;
        const float sg = 1.f / (1.f + expf(-g_ray));
        sdoff[kk * BEV_MAX_DEPTH + dd] = a.depth_base[dd] + (sg * 2.f - 1.f) * a.d_region / (float)D / 2.f;
    }
    if (r_soft) {                                      // wave-uniform
        const float mq = bev_wave_max(g_q);
        const float eq = ln < T ? expf(g_q - mq) : 0.f;
        const float sumq = bev_wave_sum(eq);
        if (ln < T)
            sq[wk * T + ln] = eq / sumq;
        if (a.B == 1) {
            const float ml = bev_wave_max(g_lg);
            const float el = ln < P ? expf(g_lg - ml) : 0.f;
            const float suml = bev_wave_sum(el);
            if (ln < P)
                sattn[wk * P + ln] = el / suml;
        }
    }
    __syncthreads();
    BEV_STAMP(1);
    // phase B: per-frame keypoints
    const int H = a.H, W = a.W;
    const unsigned pix_bytes = (unsigned)(a.heads * 64 * sizeof(FT));   // one pixel: heads x 64 channels
    for (int i = tid; i < nitems * TP; i += 256) {
        const int kk = i / TP, r = i - kk * TP, t = r / P, p = r - t * P;
        const int it = i0 + kk, q = it / a.heads, h = it % a.heads;
        float loc[2];
        if (a.B == 1) {
            bev_warp(a, sbase[(kk * P + p) * 2], sbase[(kk * P + p) * 2 + 1], svel[kk * 2], svel[kk * 2 + 1], std_[t],
                     sdoff[kk * BEV_MAX_DEPTH + p % D], loc);
        } else {
            const int fi = b * T + t;                // value frame index
            const int bq = fi % a.B, tq = fi / a.B;  // whose locations it is paired with (quirk Q2)
            bev_keypoint(a, s, bq, tq, q, h, p, loc);
            // point softmax of the paired batch b' (rare path: every thread redoes the P-term reduction)
            const float *lg = s.scale + ((size_t)bq * a.Q + q) * a.ld_scale + (size_t)h * P;
            float wmax = lg[0];
            for (int pj = 1; pj < P; ++pj)
                wmax = fmaxf(wmax, lg[pj]);
            float s2 = 0.f;
            for (int pj = 0; pj < P; ++pj)
                s2 += expf(lg[pj] - wmax);
            sattn[(kk * T + t) * P + p] = expf(lg[p] - wmax) / s2;
        }
        {
            // Tap list entry of this keypoint: bilinear footprint (Deformable-DETR semantics, align_corners=False, zero padding),
            // the point's attention weight and the frame weight folded into the four tap weights.  Computed once here
            // (one thread per keypoint) instead of by each of the 16 lanes that later gather the point.
            const float wgt = sattn[(kk * Tw + (a.B > 1 ? t : 0)) * P + p] * sq[kk * T + t];
            const float h_im = loc[1] * (float)H - 0.5f, w_im = loc[0] * (float)W - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool t_ok = in && h_low >= 0, b_ok = in && h_high <= H - 1;
            const bool l_ok = w_low >= 0, r_ok = w_high <= W - 1;
            // byte offset of (frame b*T+t, pixel 0, head h, channel 0) in the value buffer
            const unsigned fbase = ((unsigned)(b * T + t) * (unsigned)(H * W) * (unsigned)a.heads + (unsigned)h) * (unsigned)(64 * sizeof(FT));
            // GL = 16: subset = p & 3, slot = t * npp + (p >> 2) (frame-major inside a subset).  GL = 8: the item's T * P keypoints dealt
            // round-robin to its eight groups in frame-major order (subset = r & 7, slot = r >> 3 with r = t * P + p)
            static_assert(BEV_TS == 4, "point subset = p & 3, slot = p >> 2");
            float *e = GL == 16 ? stab + (((kk * TSN + (p & (BEV_TS - 1))) * Lg) + t * npp + (p >> 2)) * 8
                                : stab + (((kk * TSN + (r & (TSN - 1))) * Lg) + r / TSN) * 8;
            bev_u4 off;
            off.x = t_ok && l_ok ? fbase + (unsigned)(h_low * W + w_low) * pix_bytes : BEV_TAP_OUTSIDE;
            off.y = t_ok && r_ok ? fbase + (unsigned)(h_low * W + w_high) * pix_bytes : BEV_TAP_OUTSIDE;
            off.z = b_ok && l_ok ? fbase + (unsigned)(h_high * W + w_low) * pix_bytes : BEV_TAP_OUTSIDE;
            off.w = b_ok && r_ok ? fbase + (unsigned)(h_high * W + w_high) * pix_bytes : BEV_TAP_OUTSIDE;
            *reinterpret_cast<bev_u4 *>(e) = off;
            rac_f4 tw = {hh * hw * wgt, hh * lw * wgt, lh * hw * wgt, lh * lw * wgt};
            if (sizeof(FT) == 2 && s.vscale) {
                // int16 block storage: the scale of each tap's (pixel, head) block, folded into its weight (a tap outside the map
                // reads zeros whatever its weight)
                const float *sb = s.vscale + ((size_t)(b * T + t) * (size_t)(H * W)) * a.heads + h;
                const float s0 = t_ok && l_ok ? sb[(size_t)(h_low * W + w_low) * a.heads] : 0.f;
                const float s1 = t_ok && r_ok ? sb[(size_t)(h_low * W + w_high) * a.heads] : 0.f;
                const float s2 = b_ok && l_ok ? sb[(size_t)(h_high * W + w_low) * a.heads] : 0.f;
                const float s3 = b_ok && r_ok ? sb[(size_t)(h_high * W + w_high) * a.heads] : 0.f;
                tw.x *= s0; tw.y *= s1; tw.z *= s2; tw.w *= s3;
            }
            *reinterpret_cast<rac_f4 *>(e + 4) = tw;
        }
        if (s.loc_out) {
            float *lo = s.loc_out + (((((size_t)b * a.Q + q) * a.heads + h) * T + t) * P + p) * 2;
            lo[0] = loc[0];
            lo[1] = loc[1];
        }
    }
    __syncthreads();
    BEV_STAMP(2);

    // phase C: a group walks its list (frame-major: the chip works on (nearly) one frame at a time, which the L2s /
    // Infinity Cache hold better than all of them), BEV_U keypoints = 4 * BEV_U taps in flight.  Per tap: one add for
    // the lane's channel offset, one buffer load, two packed FMAs -- the gather runs at the L1 rate, not the VALU's.
    const bool live = k < nitems;
    if constexpr (GL == 16) {
        rac_acc4 acc4 = rac_acc4_zero();
        if (live) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.value), 0, a.value_bytes, 0x00020000);
            const unsigned lane_off = (unsigned)(c4 * 4 * sizeof(FT));
            const float *e = stab + (k * TSN + ts) * Lg * 8;          // same address for the 16 lanes of the group
            for (int j0 = 0; j0 < Lg; j0 += BEV_U) {
                rac_f4 v[BEV_U][4], tw[BEV_U];
#pragma unroll
                for (int u = 0; u < BEV_U; ++u) {
                    const bev_u4 o = *reinterpret_cast<const bev_u4 *>(e + (j0 + u) * 8);
                    tw[u] = *reinterpret_cast<const rac_f4 *>(e + (j0 + u) * 8 + 4);
                    v[u][0] = bev_tap<FT>(rsrc, o.x + lane_off);
                    v[u][1] = bev_tap<FT>(rsrc, o.y + lane_off);
                    v[u][2] = bev_tap<FT>(rsrc, o.z + lane_off);
                    v[u][3] = bev_tap<FT>(rsrc, o.w + lane_off);
                }
#pragma unroll
                for (int u = 0; u < BEV_U; ++u) {
                    const float w4[4] = {tw[u].x, tw[u].y, tw[u].z, tw[u].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        rac_tap_fma(acc4, v[u][c].x, v[u][c].y, v[u][c].z, v[u][c].w, w4[c]);
                    }
                }
            }
        }
        rac_f4 acc;
        rac_acc4_get(acc4, acc.x, acc.y, acc.z, acc.w);
        BEV_STAMP(3);
        // phase D: fixed-order sum of the four point subsets (deterministic, no atomics)
        *reinterpret_cast<rac_f4 *>(spart + (k * TSN + ts) * 64 + c4 * 4) = acc;
        __syncthreads();
        if (live && ts == 0) {
            rac_f4 o = acc;
#pragma unroll
            for (int u = 1; u < TSN; ++u) {
                const rac_f4 pz = *reinterpret_cast<const rac_f4 *>(spart + (k * TSN + u) * 64 + c4 * 4);
                o.x += pz.x; o.y += pz.y; o.z += pz.z; o.w += pz.w;
            }
            *reinterpret_cast<rac_f4 *>(s.out + ((size_t)b * per_b + (i0 + k)) * 64 + c4 * 4) = o;
        }
    } else {
        // int16 block storage, 8 lanes per tap row: one 16-byte load = this lane's 8 channels of the tap
        float accv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            accv[c] = 0.f;
        if (live) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.value), 0, a.value_bytes, 0x00020000);
            const unsigned lane_off = (unsigned)(c4 * 8 * sizeof(FT));
            const float *e = stab + (k * TSN + ts) * Lg * 8;
            for (int j0 = 0; j0 < Lg; j0 += BEV_U) {
                bev_u4 v[BEV_U][4];
                rac_f4 tw[BEV_U];
#pragma unroll
                for (int u = 0; u < BEV_U; ++u) {
                    const bev_u4 o = *reinterpret_cast<const bev_u4 *>(e + (j0 + u) * 8);
                    tw[u] = *reinterpret_cast<const rac_f4 *>(e + (j0 + u) * 8 + 4);
                    v[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + lane_off, 0, 0);
                    v[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + lane_off, 0, 0);
                    v[u][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + lane_off, 0, 0);
                    v[u][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + lane_off, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < BEV_U; ++u) {
                    const float w4[4] = {tw[u].x, tw[u].y, tw[u].z, tw[u].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned d[4] = {v[u][c].x, v[u][c].y, v[u][c].z, v[u][c].w};
#pragma unroll
                        for (int h2 = 0; h2 < 4; ++h2) {
                            accv[2 * h2] = __builtin_fmaf((float)(short)(d[h2] & 0xffffu), w4[c], accv[2 * h2]);
                            accv[2 * h2 + 1] = __builtin_fmaf((float)((int)d[h2] >> 16), w4[c], accv[2 * h2 + 1]);
                        }
                    }
                }
            }
        }
        BEV_STAMP(3);
        // phase D: fixed-order sum of the eight keypoint subsets
        *reinterpret_cast<rac_f4 *>(spart + (k * TSN + ts) * 64 + c4 * 8) = (rac_f4){accv[0], accv[1], accv[2], accv[3]};
        *reinterpret_cast<rac_f4 *>(spart + (k * TSN + ts) * 64 + c4 * 8 + 4) = (rac_f4){accv[4], accv[5], accv[6], accv[7]};
        __syncthreads();
        if (live && ts == 0) {
#pragma unroll
            for (int u = 1; u < TSN; ++u)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    accv[c] += spart[(k * TSN + u) * 64 + c4 * 8 + c];
            float *op = s.out + ((size_t)b * per_b + (i0 + k)) * 64 + c4 * 8;
            *reinterpret_cast<rac_f4 *>(op) = (rac_f4){accv[0], accv[1], accv[2], accv[3]};
            *reinterpret_cast<rac_f4 *>(op + 4) = (rac_f4){accv[4], accv[5], accv[6], accv[7]};
        }
    }
    BEV_STAMP(4);
#ifdef BEV_STAMPS
    if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < BEV_STAMP_WGS) {
        // where the workgroup ran: HW_ID (gfx9: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13) and the XCC id register
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        bev_stamp_buf[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

#ifdef BEV_STAMPS
extern "C" int rac_dbg_bev_stamps(unsigned long long *host_out, int n_wgs)
{
    if (n_wgs > BEV_STAMP_WGS)
        n_wgs = BEV_STAMP_WGS;
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(bev_stamp_buf), sizeof(unsigned long long) * 8 * n_wgs, 0, hipMemcpyDeviceToHost);
}
#endif

static int bev_launch(int nstreams, const void *const *values, const float *const *vscales, const float *const *offsets, const float *const *ray_logits,
                      const float *const *scale_logits, const float *const *queue_logits, float *const *outs, float *const *loc_outs,
                      const float *query_bbox, const float *box_table, const float *time_diff, int ld_off, int ld_ray,
                      int ld_scale, int ld_queue, int B, int T, int Q, int heads, int NP, int D, int H, int W, int dim,
                      const float *pc_range, const float *depth_base, float d_region, int dtype, void *stream)
{
    RAC_CHECK_ARG(nstreams >= 1 && nstreams <= BEV_MAX_STREAMS, "rac_bev_sampling_fwd: %d streams (1..%d)", nstreams, BEV_MAX_STREAMS);
    RAC_CHECK_ARG(dim == 64, "rac_bev_sampling_fwd: dim=%d (the fused kernel is built for 64 channels per head)", dim);
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && T >= 1 && heads >= 1 && NP >= 1 && D >= 1 && D <= BEV_MAX_DEPTH && H >= 1 && W >= 1,
                  "rac_bev_sampling_fwd: bad sizes B=%d T=%d Q=%d heads=%d NP=%d D=%d H=%d W=%d", B, T, Q, heads, NP, D, H, W);
    RAC_CHECK_ARG(dtype == RAC_F32 || dtype == RAC_BF16 || dtype == RAC_I16, "rac_bev_sampling_fwd: dtype %d", dtype);
    RAC_CHECK_ARG((dtype == RAC_I16) == (vscales != nullptr), "rac_bev_sampling_fwd: int16 value streams come with their scale tables (and only they)");
    const int P = NP * D;
    const int npp = (P + BEV_TS - 1) / BEV_TS;
    const int tsn = dtype == RAC_I16 ? 8 : BEV_TS;                       // groups per item (bev_sampling_d64_kernel's TSN)
    const int list_len = dtype == RAC_I16 ? ((T * P + 7) / 8 + BEV_U - 1) / BEV_U * BEV_U : (T * npp + BEV_U - 1) / BEV_U * BEV_U;
    const size_t lds = ((size_t)BEV_GI * tsn * list_len * 8 + (size_t)BEV_GI * tsn * 64 + (size_t)BEV_GI * (B > 1 ? T : 1) * P +
                        (size_t)BEV_GI * T + (size_t)BEV_GI * P * 2 + (size_t)BEV_GI * BEV_MAX_DEPTH + (size_t)BEV_GI * 2 + (size_t)T) * sizeof(float);
    RAC_CHECK_ARG(lds <= 64 * 1024, "rac_bev_sampling_fwd: T*P=%d too large for the LDS staging", T * P);
    const size_t value_bytes = (size_t)B * T * H * W * heads * 64 * (dtype == RAC_F32 ? 4 : 2);   // (bf16 and int16: 2 bytes)
    RAC_CHECK_ARG(value_bytes < (size_t)BEV_TAP_OUTSIDE, "rac_bev_sampling_fwd: value maps of %zu bytes (the tap offsets are 31-bit)", value_bytes);
    if (B == 0 || Q == 0)
        return 0;
    RAC_CHECK_ARG(box_table != nullptr, "rac_bev_sampling_fwd: box_table is null (run rac_box_prep_fwd first)");
    RAC_CHECK_ARG((reinterpret_cast<uintptr_t>(box_table) & 15) == 0, "rac_bev_sampling_fwd: box_table must be 16-byte aligned");
    RAC_CHECK_ARG(T <= 64, "rac_bev_sampling_fwd: T=%d frames (max 64: one lane per frame in the frame softmax)", T);
    RAC_CHECK_ARG(BEV_GI * P <= 128 && BEV_GI * D <= 64, "rac_bev_sampling_fwd: NP*D=%d (max %d) or D=%d (max %d) exceed the workgroup's staging roles", P, 128 / BEV_GI, D, 64 / BEV_GI);
    RAC_CHECK_ARG(values && offsets && ray_logits && scale_logits && queue_logits && outs && query_bbox && time_diff && pc_range && depth_base,
                  "rac_bev_sampling_fwd: null pointer");
    BevArgs a;
    for (int i = 0; i < BEV_MAX_STREAMS; ++i) {
        const int j = i < nstreams ? i : 0;
        RAC_CHECK_ARG(values[j] && offsets[j] && ray_logits[j] && scale_logits[j] && queue_logits[j] && outs[j],
                      "rac_bev_sampling_fwd: null pointer in stream %d", j);
        a.s[i].value = values[j]; a.s[i].off = offsets[j]; a.s[i].ray = ray_logits[j]; a.s[i].scale = scale_logits[j];
        a.s[i].queue = queue_logits[j]; a.s[i].out = outs[j]; a.s[i].loc_out = loc_outs ? loc_outs[j] : nullptr;
        a.s[i].vscale = vscales ? vscales[j] : nullptr;
        RAC_CHECK_ARG(!vscales || vscales[j], "rac_bev_sampling_fwd: null scale table in stream %d", j);
    }
    a.qbox = query_bbox; a.box = box_table; a.time_diff = time_diff;
    for (int i = 0; i < BEV_MAX_DEPTH; ++i)
        a.depth_base[i] = i < D ? depth_base[i] : 0.f;
    for (int i = 0; i < 6; ++i)
        a.pc[i] = pc_range[i];
    a.d_region = d_region;
    a.B = B; a.T = T; a.Q = Q; a.heads = heads; a.NP = NP; a.D = D; a.P = P; a.H = H; a.W = W;
    a.ld_off = ld_off; a.ld_ray = ld_ray; a.ld_scale = ld_scale; a.ld_queue = ld_queue;
    a.blocks_per_b = (Q * heads + BEV_GI - 1) / BEV_GI;
    a.value_bytes = (unsigned)value_bytes;
    a.list_len = list_len;
    a.list_holes = dtype == RAC_I16 ? (list_len * 8 != T * P ? 1 : 0) : ((P % BEV_TS != 0 || list_len != T * npp) ? 1 : 0);
    a.xcd_remap = 1;
    const dim3 grid(B * a.blocks_per_b, nstreams);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RAC_F32)
        hipLaunchKernelGGL((bev_sampling_d64_kernel<float, 16>), grid, dim3(256), lds, st, a);
    else if (dtype == RAC_I16)
        hipLaunchKernelGGL((bev_sampling_d64_kernel<short, 8>), grid, dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL((bev_sampling_d64_kernel<unsigned short, 16>), grid, dim3(256), lds, st, a);
    return rac_launch_status("rac_bev_sampling_fwd");
}

extern "C" int rac_bev_sampling_fwd(const void *value, const float *query_bbox, const float *box_table,
                                    const float *offsets,
                                    const float *ray_logits, const float *scale_logits, const float *queue_logits,
                                    const float *time_diff, float *out, float *loc_out, int ld_off, int ld_ray,
                                    int ld_scale, int ld_queue, int B, int T, int Q, int heads,
                                    int NP, int D, int H, int W, int dim, const float *pc_range,
                                    const float *depth_base, float d_region, int dtype, void *stream)
{
    return bev_launch(1, &value, nullptr, &offsets, &ray_logits, &scale_logits, &queue_logits, &out, &loc_out, query_bbox, box_table, time_diff,
                      ld_off, ld_ray, ld_scale, ld_queue, B, T, Q, heads, NP, D, H, W, dim, pc_range, depth_base, d_region, dtype, stream);
}

extern "C" int rac_bev_sampling_multi_fwd(int nstreams, const void *const *values, const float *const *offsets,
                                          const float *const *ray_logits, const float *const *scale_logits,
                                          const float *const *queue_logits, float *const *outs, const float *query_bbox,
                                          const float *box_table, const float *time_diff, int ld_off, int ld_ray, int ld_scale,
                                          int ld_queue, int B, int T, int Q, int heads, int NP, int D, int H, int W, int dim,
                                          const float *pc_range, const float *depth_base, float d_region, int dtype, void *stream)
{
    return bev_launch(nstreams, values, nullptr, offsets, ray_logits, scale_logits, queue_logits, outs, nullptr, query_bbox, box_table, time_diff,
                      ld_off, ld_ray, ld_scale, ld_queue, B, T, Q, heads, NP, D, H, W, dim, pc_range, depth_base, d_region, dtype, stream);
}

extern "C" int rac_bev_sampling_multi_q16_fwd(int nstreams, const void *const *values, const float *const *value_scales,
                                              const float *const *offsets, const float *const *ray_logits,
                                              const float *const *scale_logits, const float *const *queue_logits, float *const *outs,
                                              const float *query_bbox, const float *box_table, const float *time_diff, int ld_off,
                                              int ld_ray, int ld_scale, int ld_queue, int B, int T, int Q, int heads, int NP, int D,
                                              int H, int W, int dim, const float *pc_range, const float *depth_base, float d_region,
                                              void *stream)
{
    RAC_CHECK_ARG(value_scales != nullptr, "rac_bev_sampling_multi_q16_fwd: null scale tables");
    return bev_launch(nstreams, values, value_scales, offsets, ray_logits, scale_logits, queue_logits, outs, nullptr, query_bbox, box_table,
                      time_diff, ld_off, ld_ray, ld_scale, ld_queue, B, T, Q, heads, NP, D, H, W, dim, pc_range, depth_base, d_region,
                      RAC_I16, stream);
}
