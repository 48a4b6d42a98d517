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
    int xcd_remap;           // 1: blocks that share an XCD (blockIdx & 7) take a contiguous range of items (speed only)
};

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
    float dist = sqrtf(ex * ex + ey * ey) / 65.0f;
    const float th = fmodf(atan2f(ey, ex) + BEV_TWO_PI, BEV_TWO_PI) / BEV_TWO_PI;
    const int dd = p % a.D;
    const float sg = 1.f / (1.f + expf(-s.ray[((size_t)bq * a.Q + q) * a.ld_ray + dd]));
    dist += a.depth_base[dd] + (sg * 2.f - 1.f) * a.d_region / (float)a.D / 2.f;
    const float ang = th * BEV_TWO_PI, rad = dist * 65.0f;
    loc2[0] = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
    loc2[1] = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
}

template <typename FT>
__device__ __forceinline__ rac_f4 bev_tap(const FT *base, long pix, int stride, bool ok)
{
    rac_f4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        v = rac_ld4(base + pix * stride);
    return v;
}

#define BEV_GI 4   /* items (b,q,head) per workgroup */
#define BEV_TS 4   /* point subsets per item: 16-lane group (k, ts) handles points ts, ts+4, ... of every frame */

// per-(t,p) half of the keypoint chain for B==1: warp the T-invariant base point, polar jitter.
__device__ __forceinline__ void bev_warp(const BevArgs &a, float px, float py, float vx, float vy, float td,
                                         float doff, float *loc2)
{
    const float sx = a.pc[3] - a.pc[0], sy = a.pc[4] - a.pc[1];
    px -= vx * td;
    py -= vy * td;
    const float nx = (px - a.pc[0]) / sx, ny = (py - a.pc[1]) / sy;
    const float ex = nx * 102.4f - 51.2f, ey = ny * 102.4f - 51.2f;
    const float dist = sqrtf(ex * ex + ey * ey) / 65.0f + doff;
    const float th = fmodf(atan2f(ey, ex) + BEV_TWO_PI, BEV_TWO_PI) / BEV_TWO_PI;
    const float ang = th * BEV_TWO_PI, rad = dist * 65.0f;
    loc2[0] = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
    loc2[1] = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
}

template <typename FT>
__global__ __launch_bounds__(256, 4) void bev_sampling_d64_kernel(const BevArgs a)
{
    extern __shared__ float smem[];
    const BevStream &s = a.s[blockIdx.y];
    const int tid = threadIdx.x;
    const int c4 = tid & 15, grp = tid >> 4;
    const int k = grp >> 2, ts = grp & 3;       // item within the workgroup, frame subset
    const int T = a.T, P = a.P, TP = a.T * a.P, D = a.D;

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

    float *sloc = smem;                        // [GI][T][P][2]
    const int Tw = a.B > 1 ? T : 1;            // B > 1: the point weights depend on the frame (paired batch, quirk Q2)
    float *sattn = sloc + BEV_GI * TP * 2;     // [GI][Tw][P]
    float *sq = sattn + BEV_GI * Tw * P;       // [GI][T]
    float *sbase = sq + BEV_GI * T;            // [GI][P][2]  T-invariant base points (B==1)
    float *sdoff = sbase + BEV_GI * P * 2;     // [GI][D]
    float *spart = sdoff + BEV_GI * BEV_MAX_DEPTH;  // [GI][TS][64] partial sums
    float *stab = spart + BEV_GI * BEV_TS * 64;     // [GI][T][P][8]: 4 tap pixel indices (int, -1 = outside) + 4 tap weights

    // phase A: T-invariant pieces.  threads [0, GI*P): base points; [128,128+GI*D): depth offsets;
    // [192,192+GI): softmaxes of the point weights and of the frame weights.
    if (tid < nitems * P && a.B == 1) {
        const int kk = tid / P, p = tid - kk * P;
        const int it = i0 + kk, q = it / a.heads, h = it % a.heads;
        const float *bt = a.box + ((size_t)b * a.Q + q) * 8;
        const float *o = s.off + ((size_t)b * a.Q + q) * a.ld_off + ((size_t)h * P + p) * 2;
        const float dx = bt[3] * o[0], dy = bt[4] * o[1];
        sbase[tid * 2] = bt[0] + (dx * bt[6] - dy * bt[7]);
        sbase[tid * 2 + 1] = bt[1] + (dx * bt[7] + dy * bt[6]);
    }
    if (tid >= 128 && tid < 128 + nitems * D && a.B == 1) {
        const int kk = (tid - 128) / D, dd = (tid - 128) - kk * D;
        const int q = (i0 + kk) / a.heads;
        const float sg = 1.f / (1.f + expf(-s.ray[((size_t)b * a.Q + q) * a.ld_ray + dd]));
        sdoff[kk * BEV_MAX_DEPTH + dd] = a.depth_base[dd] + (sg * 2.f - 1.f) * a.d_region / (float)D / 2.f;
    }
    if (tid >= 192 && tid < 192 + nitems) {
        const int kk = tid - 192;
        const int it = i0 + kk, q = it / a.heads, h = it % a.heads;
        const float *qg = s.queue + ((size_t)b * a.Q + q) * a.ld_queue;
        float mx = qg[0];
        for (int t = 1; t < T; ++t)
            mx = fmaxf(mx, qg[t]);
        float sum = 0.f;
        for (int t = 0; t < T; ++t) {
            const float e = expf(qg[t] - mx);
            sq[kk * T + t] = e;
            sum += e;
        }
        for (int t = 0; t < T; ++t)
            sq[kk * T + t] /= sum;
        if (a.B == 1) {
            const float *lg = s.scale + ((size_t)b * a.Q + q) * a.ld_scale + (size_t)h * P;
            float m2 = lg[0];
            for (int p = 1; p < P; ++p)
                m2 = fmaxf(m2, lg[p]);
            float s2 = 0.f;
            for (int p = 0; p < P; ++p) {
                const float e = expf(lg[p] - m2);
                sattn[kk * P + p] = e;
                s2 += e;
            }
            for (int p = 0; p < P; ++p)
                sattn[kk * P + p] /= s2;
        }
    }
    __syncthreads();
    // phase B: per-frame keypoints
    for (int i = tid; i < nitems * TP; i += 256) {
        const int kk = i / TP, r = i - kk * TP, t = r / P, p = r - t * P;
        const int it = i0 + kk, q = it / a.heads, h = it % a.heads;
        if (a.B == 1) {
            const float *qb = a.qbox + ((size_t)b * a.Q + q) * 10;
            bev_warp(a, sbase[(kk * P + p) * 2], sbase[(kk * P + p) * 2 + 1], qb[8], qb[9], a.time_diff[b * T + t],
                     sdoff[kk * BEV_MAX_DEPTH + p % D], sloc + i * 2);
        } else {
            const int fi = b * T + t;                // value frame index
            const int bq = fi % a.B, tq = fi / a.B;  // whose locations it is paired with (quirk Q2)
            bev_keypoint(a, s, bq, tq, q, h, p, sloc + i * 2);
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
            // Tap table of this keypoint: bilinear footprint (Deformable-DETR semantics, align_corners=False, zero padding),
            // the point's attention weight and the frame weight folded into the four tap weights.  Computed once here
            // (one thread per keypoint) instead of by each of the 16 lanes that later gather the point.
            const float wgt = sattn[(kk * Tw + (a.B > 1 ? t : 0)) * P + p] * sq[kk * T + t];
            const int Hh = a.H, Ww = a.W;
            const float h_im = sloc[i * 2 + 1] * (float)Hh - 0.5f, w_im = sloc[i * 2] * (float)Ww - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)Hh && w_im < (float)Ww;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool t_ok = in && h_low >= 0, b_ok = in && h_high <= Hh - 1;
            const bool l_ok = w_low >= 0, r_ok = w_high <= Ww - 1;
            int *ti = reinterpret_cast<int *>(stab + i * 8);
            ti[0] = t_ok && l_ok ? h_low * Ww + w_low : -1;
            ti[1] = t_ok && r_ok ? h_low * Ww + w_high : -1;
            ti[2] = b_ok && l_ok ? h_high * Ww + w_low : -1;
            ti[3] = b_ok && r_ok ? h_high * Ww + w_high : -1;
            stab[i * 8 + 4] = hh * hw * wgt;
            stab[i * 8 + 5] = hh * lw * wgt;
            stab[i * 8 + 6] = lh * hw * wgt;
            stab[i * 8 + 7] = lh * lw * wgt;
        }
        if (s.loc_out) {
            float *lo = s.loc_out + (((((size_t)b * a.Q + q) * a.heads + h) * T + t) * P + p) * 2;
            lo[0] = sloc[i * 2];
            lo[1] = sloc[i * 2 + 1];
        }
    }
    __syncthreads();

    rac_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool live = k < nitems;
    const int it = i0 + (live ? k : 0), h = it % a.heads;
    const int H = a.H, W = a.W, stride = a.heads * 64;
    const long keys = (long)H * W;
    if (live) {
        // A group owns the points p = ts, ts+4, ... of every frame and walks its (frame, point) pairs frame-major, four
        // at a time (16 taps in flight, every batch full): all groups move through the frames in the same order, so the
        // chip works on (nearly) one 16.8 MB frame at a time, which the L2s / Infinity Cache hold better than four.
        const int npp = (P - ts + BEV_TS - 1) / BEV_TS;   // this subset's points per frame
        const int total = T * npp;
        for (int j0 = 0; j0 < total; j0 += 4) {
            rac_f4 v[4][4];
            float tw[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                const bool act = j < total;
                const int jj = act ? j : total - 1;
                const int t = jj / npp, p = ts + BEV_TS * (jj - t * npp);
                const FT *base = (const FT *)s.value + ((size_t)(b * T + t) * keys * a.heads + h) * 64 + c4 * 4;
                const float *e = stab + (k * TP + t * P + p) * 8;          // same address for the 16 lanes of the group
                const rac_f4 ei = *reinterpret_cast<const rac_f4 *>(e), ew = *reinterpret_cast<const rac_f4 *>(e + 4);
                const int o0 = __float_as_int(ei.x), o1 = __float_as_int(ei.y), o2 = __float_as_int(ei.z), o3 = __float_as_int(ei.w);
                v[u][0] = bev_tap(base, (long)o0, stride, act && o0 >= 0);
                v[u][1] = bev_tap(base, (long)o1, stride, act && o1 >= 0);
                v[u][2] = bev_tap(base, (long)o2, stride, act && o2 >= 0);
                v[u][3] = bev_tap(base, (long)o3, stride, act && o3 >= 0);
                tw[u][0] = act ? ew.x : 0.f;
                tw[u][1] = act ? ew.y : 0.f;
                tw[u][2] = act ? ew.z : 0.f;
                tw[u][3] = act ? ew.w : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x += tw[u][0] * v[u][0].x + tw[u][1] * v[u][1].x + tw[u][2] * v[u][2].x + tw[u][3] * v[u][3].x;
                acc.y += tw[u][0] * v[u][0].y + tw[u][1] * v[u][1].y + tw[u][2] * v[u][2].y + tw[u][3] * v[u][3].y;
                acc.z += tw[u][0] * v[u][0].z + tw[u][1] * v[u][1].z + tw[u][2] * v[u][2].z + tw[u][3] * v[u][3].z;
                acc.w += tw[u][0] * v[u][0].w + tw[u][1] * v[u][1].w + tw[u][2] * v[u][2].w + tw[u][3] * v[u][3].w;
            }
        }
    }
    // phase D: fixed-order sum of the four point subsets (deterministic, no atomics)
    *reinterpret_cast<rac_f4 *>(spart + (k * BEV_TS + ts) * 64 + c4 * 4) = acc;
    __syncthreads();
    if (live && ts == 0) {
        rac_f4 o = acc;
#pragma unroll
        for (int u = 1; u < BEV_TS; ++u) {
            const rac_f4 pz = *reinterpret_cast<const rac_f4 *>(spart + (k * BEV_TS + u) * 64 + c4 * 4);
            o.x += pz.x; o.y += pz.y; o.z += pz.z; o.w += pz.w;
        }
        *reinterpret_cast<rac_f4 *>(s.out + ((size_t)b * per_b + it) * 64 + c4 * 4) = o;
    }
}

static int bev_launch(int nstreams, const void *const *values, const float *const *offsets, const float *const *ray_logits,
                      const float *const *scale_logits, const float *const *queue_logits, float *const *outs, float *const *loc_outs,
                      const float *query_bbox, const float *box_table, const float *time_diff, int ld_off, int ld_ray,
                      int ld_scale, int ld_queue, int B, int T, int Q, int heads, int NP, int D, int H, int W, int dim,
                      const float *pc_range, const float *depth_base, float d_region, int dtype, void *stream)
{
    RAC_CHECK_ARG(nstreams >= 1 && nstreams <= BEV_MAX_STREAMS, "rac_bev_sampling_fwd: %d streams (1..%d)", nstreams, BEV_MAX_STREAMS);
    RAC_CHECK_ARG(dim == 64, "rac_bev_sampling_fwd: dim=%d (the fused kernel is built for 64 channels per head)", dim);
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && T >= 1 && heads >= 1 && NP >= 1 && D >= 1 && D <= BEV_MAX_DEPTH && H >= 1 && W >= 1,
                  "rac_bev_sampling_fwd: bad sizes B=%d T=%d Q=%d heads=%d NP=%d D=%d H=%d W=%d", B, T, Q, heads, NP, D, H, W);
    RAC_CHECK_ARG(dtype == RAC_F32 || dtype == RAC_BF16, "rac_bev_sampling_fwd: dtype %d", dtype);
    const int P = NP * D;
    const size_t lds = ((size_t)BEV_GI * T * P * 2 + (size_t)BEV_GI * (B > 1 ? T : 1) * P + (size_t)BEV_GI * T + (size_t)BEV_GI * P * 2 +
                        (size_t)BEV_GI * BEV_MAX_DEPTH + (size_t)BEV_GI * BEV_TS * 64 + (size_t)BEV_GI * T * P * 8) * sizeof(float);
    RAC_CHECK_ARG(lds <= 64 * 1024, "rac_bev_sampling_fwd: T*P=%d too large for the LDS staging", T * P);
    if (B == 0 || Q == 0)
        return 0;
    RAC_CHECK_ARG(box_table != nullptr, "rac_bev_sampling_fwd: box_table is null (run rac_box_prep_fwd first)");
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
    a.xcd_remap = 1;
    const dim3 grid(B * a.blocks_per_b, nstreams);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RAC_F32)
        hipLaunchKernelGGL(bev_sampling_d64_kernel<float>, grid, dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL(bev_sampling_d64_kernel<unsigned short>, grid, dim3(256), lds, st, a);
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
    return bev_launch(1, &value, &offsets, &ray_logits, &scale_logits, &queue_logits, &out, &loc_out, query_bbox, box_table, time_diff,
                      ld_off, ld_ray, ld_scale, ld_queue, B, T, Q, heads, NP, D, H, W, dim, pc_range, depth_base, d_region, dtype, stream);
}

extern "C" int rac_bev_sampling_multi_fwd(int nstreams, const void *const *values, const float *const *offsets,
                                          const float *const *ray_logits, const float *const *scale_logits,
                                          const float *const *queue_logits, float *const *outs, const float *query_bbox,
                                          const float *box_table, const float *time_diff, int ld_off, int ld_ray, int ld_scale,
                                          int ld_queue, int B, int T, int Q, int heads, int NP, int D, int H, int W, int dim,
                                          const float *pc_range, const float *depth_base, float d_region, int dtype, void *stream)
{
    return bev_launch(nstreams, values, offsets, ray_logits, scale_logits, queue_logits, outs, nullptr, query_bbox, box_table, time_diff,
                      ld_off, ld_ray, ld_scale, ld_queue, B, T, Q, heads, NP, D, H, W, dim, pc_range, depth_base, d_region, dtype, stream);
}
