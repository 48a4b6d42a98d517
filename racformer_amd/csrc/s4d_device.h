// s4d_device.h -- the keypoint / projection / tap arithmetic of the adaptive 4D sampling, shared by the stand-alone fused
// sampling kernel (sampling_fused.hip) and the sampling + AdaptiveMixing kernel (mixing.hip): ONE definition, so that both
// compute a sampling point, its camera choice and its bilinear taps with the same instructions in the same order.
//   RaCFormerSampling.inner_forward  models/racformer_transformer.py:361-408  (keypoints)
//   sampling_4d                      models/sparsebev_sampling.py:45-131      (projection, validity, first-valid-view selection)
//   msmv op                          models/csrc/msmv_sampling/msmv_sampling_forward.cu:75-164   (bilinear footprint)
#pragma once
#include "rac_common.h"

#define S4D_MAX_DEPTH 16
#define S4D_MAX_CAMS 16

struct S4dArgs {
    const void *feat[RAC_MAX_LEVELS];
    int H[RAC_MAX_LEVELS];
    int W[RAC_MAX_LEVELS];
    unsigned feat_bytes[RAC_MAX_LEVELS];  // size of one slot's N maps of each level (the buffer descriptors' ranges)
    const float *qbox;       // [B,Q,10]
    const float *box;        // [B,Q,8] from rac_box_prep_fwd (cx,cy,cz,w,l,h,cos,sin)
    const float *off;        // [B,Q,G*P*3]
    const float *ray;        // [B,Q,D]
    const float *scale;      // [B,Q,G,T,P,L] logits
    const float *time_diff;  // [B,T]
    const float *l2i;        // [B,T*N,16]
    float *out;              // [B,Q,G,T*P,C]
    float *loc_out;          // optional [S,Q,P,3]
    float *w_out;            // optional [S,Q,P,L]
    const unsigned char *view_in;  // optional [S,Q,P]: camera index to use instead of the first valid one (parity tests)
    float depth_base[S4D_MAX_DEPTH];
    float pc[6];
    float d_region, image_h, image_w, eps;
    int L, B, T, N, G, Q, NP, D, P;
    int ld_off, ld_ray, ld_scale;  // row strides (floats) of off / ray / scale: slices of one fused GEMM output
    int blocks_per_slot;
    int rows;  // queries per workgroup (S4D_ROWS unless overridden for experiments)
};

#define S4D_TWO_PI 6.283185307179586f

template <int L>
__device__ __forceinline__ void s4d_keypoint(const S4dArgs &a, const float *sl2i, int b, int t, int g, int q,
                                             int p, float *loc3, float *wl)
{
    const float *qb = a.qbox + ((size_t)b * a.Q + q) * 10;
    const float sx = a.pc[3] - a.pc[0], sy = a.pc[4] - a.pc[1];
    // per-query constants (decode_bbox(theta_d2xy(box))) come from the box table
    const float *bt = a.box + ((size_t)b * a.Q + q) * 8;
    const float cx = bt[0], cy = bt[1], cz = bt[2];
    const float cs = bt[6], sn = bt[7];
    // make_sample_points: xyz + R_z(yaw) (wlh * offset)
    const float *o = a.off + ((size_t)b * a.Q + q) * a.ld_off + ((size_t)g * a.P + p) * 3;
    const float dx = bt[3] * o[0], dy = bt[4] * o[1], dz = bt[5] * o[2];
    float px = cx + (dx * cs - dy * sn);
    float py = cy + (dx * sn + dy * cs);
    const float pz = cz + dz;
    // velocity warp to frame t
    const float td = a.time_diff[b * a.T + t];
    px -= qb[8] * td;
    py -= qb[9] * td;
    // normalise, to polar, jitter the range, back to metric
    const float nx = (px - a.pc[0]) / sx, ny = (py - a.pc[1]) / sy;
    const float ex = nx * 102.4f - 51.2f, ey = ny * 102.4f - 51.2f;
    float dist = sqrtf(ex * ex + ey * ey) / 65.0f;
    float th = fmodf(atan2f(ey, ex) + S4D_TWO_PI, S4D_TWO_PI) / S4D_TWO_PI;
    const int dd = p % a.D;
    const float sg = 1.f / (1.f + expf(-a.ray[((size_t)b * a.Q + q) * a.ld_ray + dd]));
    dist += a.depth_base[dd] + (sg * 2.f - 1.f) * a.d_region / (float)a.D / 2.f;
    const float ang = th * S4D_TWO_PI, rad = dist * 65.0f;
    const float X = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f) * sx + a.pc[0];
    const float Y = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f) * sy + a.pc[1];
    // project into the N cameras of frame t; first valid view (0 if none)
    float u_sel = 0.f, v_sel = 0.f;
    int view = 0, own = 0;
    bool found = false;
    // view_in imposes the camera choice (the one discontinuous step of the path) from outside: with the reference's
    // own choices the whole decoder is a continuous function of its inputs, which is what the parity tests compare
    const int forced = a.view_in ? (int)a.view_in[((((size_t)b * a.T + t) * a.G + g) * a.Q + q) * a.P + p] : -1;
    for (int n = 0; n < a.N; ++n) {
        const float *m = sl2i + n * 16;
        const float camx = m[0] * X + m[1] * Y + m[2] * pz + m[3];
        const float camy = m[4] * X + m[5] * Y + m[6] * pz + m[7];
        const float homo = m[8] * X + m[9] * Y + m[10] * pz + m[11];
        const float hz = fmaxf(homo, a.eps);
        const float u = camx / hz / a.image_w;
        const float v = camy / hz / a.image_h;
        const bool valid = homo > a.eps && v > 0.f && v < 1.f && u > 0.f && u < 1.f;
        if (forced >= 0 ? n == forced : (n == 0 || (valid && !found))) {
            u_sel = u;
            v_sel = v;
            view = n;
        }
        if (valid && !found)
            own = n;
        found = found || valid;
    }
    loc3[0] = u_sel;
    loc3[1] = v_sel;
    // integer view index (the reference stores view/(N-1) and rounds it back); the kernel's OWN choice rides in the
    // upper bits so that loc_out can report it when view_in imposes another one
    loc3[2] = (float)(view + 256 * own);
    // softmax over levels of the (b, g', t') weight slot
    const int sp = t * a.G + g;
    const int gq = sp / a.T, tq = sp % a.T;
    const float *lg = a.scale + ((size_t)b * a.Q + q) * a.ld_scale + ((((size_t)gq * a.T + tq) * a.P + p)) * L;
    float mx = lg[0];
#pragma unroll
    for (int l = 1; l < L; ++l)
        mx = fmaxf(mx, lg[l]);
    float e[L], sum = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        e[l] = expf(lg[l] - mx);
        sum += e[l];
    }
#pragma unroll
    for (int l = 0; l < L; ++l)
        wl[l] = e[l] / sum;
}

#define S4D_TAP_OUTSIDE 0x80000000u   /* tap offset past the end of a level's buffer: the buffer load returns zeros */
typedef float s4d_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int s4d_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int s4d_u4 __attribute__((ext_vector_type(4)));

// Four channels of one tap through the level's buffer descriptor: its range check stands in for the branches of the
// bilinear footprint (a tap outside the map carries the offset S4D_TAP_OUTSIDE and reads as zero).
template <typename FT>
__device__ __forceinline__ rac_f4 s4d_tap(__amdgpu_buffer_rsrc_t rsrc, unsigned off);
template <>
__device__ __forceinline__ rac_f4 s4d_tap<float>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    return __builtin_bit_cast(rac_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
}
template <>
__device__ __forceinline__ rac_f4 s4d_tap<unsigned short>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    const s4d_u2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);    // 4 x bf16
    return (rac_f4){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                    __uint_as_float(r.y & 0xffff0000u)};
}


// The bilinear footprint of one keypoint in level l of its camera's map, as a tap-table entry: four byte offsets (relative to
// `slot_base`, the start of the keypoint's slot inside the buffer the caller's descriptor covers; S4D_TAP_OUTSIDE = no tap)
// and the four bilinear weights with the level weight folded in.  Returns whether any tap is inside the map.
template <typename FT>
__device__ __forceinline__ bool s4d_taps_of_level(int H, int W, float lu, float lv, int view, float wl, unsigned slot_base, float *e)
{
    // (rounded statement by statement, no FP contraction: the function is inlined into two kernels whose results have to be the
    //  same bits, and whether `lv * (H - 1) - floor(.)` becomes one fma is otherwise the inliner's context-dependent choice)
#pragma clang fp contract(off)
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
    const unsigned mbase = slot_base + (unsigned)view * (unsigned)(H * W) * pix_bytes;   // camera's map inside the slot's block of the level
    s4d_u4 off;
    off.x = t_ok && l_ok ? mbase + (unsigned)(h_low * W + w_low) * pix_bytes : S4D_TAP_OUTSIDE;
    off.y = t_ok && r_ok ? mbase + (unsigned)(h_low * W + w_high) * pix_bytes : S4D_TAP_OUTSIDE;
    off.z = b_ok && l_ok ? mbase + (unsigned)(h_high * W + w_low) * pix_bytes : S4D_TAP_OUTSIDE;
    off.w = b_ok && r_ok ? mbase + (unsigned)(h_high * W + w_high) * pix_bytes : S4D_TAP_OUTSIDE;
    *reinterpret_cast<s4d_u4 *>(e) = off;
    *reinterpret_cast<rac_f4 *>(e + 4) = (rac_f4){hh * hw * wl, hh * lw * wl, lh * hw * wl, lh * lw * wl};
    return (t_ok || b_ok) && (l_ok || r_ok);
}
