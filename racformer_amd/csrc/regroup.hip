// regroup.hip -- FPN pyramid regroup to the sampling layout, for gfx950 (MI355X).
//
// Replaces the reshape/permute/.contiguous() of RaCFormerTransformerDecoder.forward
// (models/racformer_transformer.py:112-124, channel-last branch):
//     in  [B, T*N, G*C, H, W]  ->  out [B*T*G, N, H, W, C]
// i.e. for every (b,t,n,g) a [C][H*W] -> [H*W][C] transpose.  HBM-bound (reads and writes the
// whole 735 MB pyramid once per forward): 64x64 tiles through LDS (row stride 65 dwords, so the
// transposed read is bank-conflict free), 256-byte coalesced segments on both sides.
#include "rac_common.h"

template <typename OT>
__device__ __forceinline__ void regroup_store(OT *p, float v);
template <>
__device__ __forceinline__ void regroup_store<float>(float *p, float v) { *p = v; }
template <>
__device__ __forceinline__ void regroup_store<unsigned short>(unsigned short *p, float v)
{
    // round-to-nearest-even f32 -> bf16 (NaN stays NaN via the quiet bit)
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) {
        *p = (unsigned short)((u >> 16) | 0x40);
        return;
    }
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}

template <typename OT>
__global__ __launch_bounds__(256) void regroup_kernel(const float *__restrict__ in, OT *__restrict__ out,
                                                      int T, int N, int G, int C, int HW)
{
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const int hw0 = blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    int z = blockIdx.z;  // ((b*T+t)*N+n)*G+g
    const int g = z % G; z /= G;
    const int n = z % N; z /= N;
    const int t = z % T;
    const int b = z / T;
    const float *src = in + ((((size_t)b * T + t) * N + n) * G + g) * (size_t)C * HW;
    OT *dst = out + ((((size_t)b * T + t) * G + g) * N + n) * (size_t)HW * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = c0 + ty * 16 + r, hw = hw0 + tx;
        if (c < C && hw < HW)
            tile[ty * 16 + r][tx] = src[(size_t)c * HW + hw];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int hw = hw0 + ty * 16 + r, c = c0 + tx;
        if (c < C && hw < HW)
            regroup_store<OT>(dst + (size_t)hw * C + c, tile[tx][ty * 16 + r]);
    }
}

// ---- all levels of the pyramid in one launch, 16-byte accesses on both sides ---------------------------------------------
// A workgroup transposes one 64 (channels) x 64 (pixels) tile: every thread loads a 4 x 4 block (four float4 rows, the 16
// lanes of a row segment cover 256 contiguous bytes), transposes it in registers, parks it in LDS ([pixel][channel], padded
// rows) and the workgroup writes the tile back with the channels contiguous (float4 per lane, 256-byte segments).  The
// four levels share the launch (grid = sum of their tiles): the small ones no longer pay a launch each.
#define RG_MAX_LEVELS 8
struct RegroupLevel {
    const float *in;
    void *out;
    int HW;          // pixels per map
    int tiles_hw;    // ceil(HW / 64)
    int tile0;       // first workgroup of this level
};
struct RegroupArgs {
    RegroupLevel lv[RG_MAX_LEVELS];
    int L, T, N, G, C, tiles_c;
    long nz;         // B*T*N*G
};

template <typename OT>
__device__ __forceinline__ void regroup_store4(OT *p, const rac_f4 v);
template <>
__device__ __forceinline__ void regroup_store4<float>(float *p, const rac_f4 v) { rac_st4_stream(p, v); }
template <>
__device__ __forceinline__ void regroup_store4<unsigned short>(unsigned short *p, const rac_f4 v)
{
    regroup_store<unsigned short>(p, v.x);
    regroup_store<unsigned short>(p + 1, v.y);
    regroup_store<unsigned short>(p + 2, v.z);
    regroup_store<unsigned short>(p + 3, v.w);
}

template <typename OT>
__global__ __launch_bounds__(256) void regroup_multi_kernel(const RegroupArgs a)
{
    __shared__ float tile[64][68];                       // [pixel][channel], row stride 68 floats (16-byte aligned rows)
    int l = 0;
#pragma unroll
    for (int i = 1; i < RG_MAX_LEVELS; ++i)
        if (i < a.L && (int)blockIdx.x >= a.lv[i].tile0)
            l = i;
    const RegroupLevel &lv = a.lv[l];
    int r = blockIdx.x - lv.tile0;
    const int thw = r % lv.tiles_hw; r /= lv.tiles_hw;
    const int tc = r % a.tiles_c;
    long z = r / a.tiles_c;                              // ((b*T+t)*N+n)*G+g
    const int g = (int)(z % a.G); z /= a.G;
    const int n = (int)(z % a.N); z /= a.N;
    const int t = (int)(z % a.T);
    const long b = z / a.T;
    const int HW = lv.HW, C = a.C;
    const float *src = lv.in + ((((size_t)b * a.T + t) * a.N + n) * a.G + g) * (size_t)C * HW;
    OT *dst = reinterpret_cast<OT *>(lv.out) + ((((size_t)b * a.T + t) * a.G + g) * a.N + n) * (size_t)HW * C;
    const int hw0 = thw * 64, c0 = tc * 64;
    // load: thread (cq, hq) = 4 channels x 4 pixels; hq fastest over the lanes
    const int hq = threadIdx.x & 15, cq = threadIdx.x >> 4;
    rac_f4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + 4 * cq + i, hw = hw0 + 4 * hq;
        v[i] = (rac_f4){0.f, 0.f, 0.f, 0.f};
        if (c < C && hw < HW)                            // (HW % 4 == 0: a float4 never straddles the end)
            v[i] = rac_ld4_stream(src + (size_t)c * HW + hw);
    }
    // 4 x 4 register transpose -> LDS rows = pixels
    *reinterpret_cast<rac_f4 *>(&tile[4 * hq + 0][4 * cq]) = (rac_f4){v[0].x, v[1].x, v[2].x, v[3].x};
    *reinterpret_cast<rac_f4 *>(&tile[4 * hq + 1][4 * cq]) = (rac_f4){v[0].y, v[1].y, v[2].y, v[3].y};
    *reinterpret_cast<rac_f4 *>(&tile[4 * hq + 2][4 * cq]) = (rac_f4){v[0].z, v[1].z, v[2].z, v[3].z};
    *reinterpret_cast<rac_f4 *>(&tile[4 * hq + 3][4 * cq]) = (rac_f4){v[0].w, v[1].w, v[2].w, v[3].w};
    __syncthreads();
    // store: thread (pixel row p, channel quad cw); cw fastest over the lanes: 16 lanes write 256 contiguous bytes
    const int cw = threadIdx.x & 15, p0 = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + 16 * i, hw = hw0 + p, c = c0 + 4 * cw;
        if (hw < HW && c < C)
            regroup_store4<OT>(dst + (size_t)hw * C + c, *reinterpret_cast<const rac_f4 *>(&tile[p][4 * cw]));
    }
}

extern "C" int rac_regroup_multi_fwd(int L, const float *const *ins, void *const *outs, const int32_t *hw, int B, int T, int N,
                                     int G, int C, int out_dtype, void *stream)
{
    RAC_CHECK_ARG(ins && outs && hw && L >= 1 && L <= RG_MAX_LEVELS, "rac_regroup_multi_fwd: L=%d", L);
    RAC_CHECK_ARG(B >= 0 && T >= 1 && N >= 1 && G >= 1 && C >= 4 && C % 4 == 0, "rac_regroup_multi_fwd: bad sizes (C %% 4 must be 0)");
    RAC_CHECK_ARG(out_dtype == RAC_F32 || out_dtype == RAC_BF16, "rac_regroup_multi_fwd: dtype %d", out_dtype);
    RegroupArgs a;
    a.L = L; a.T = T; a.N = N; a.G = G; a.C = C; a.tiles_c = (C + 63) / 64;
    a.nz = (long)B * T * N * G;
    if (a.nz == 0)
        return 0;
    long total = 0;
    for (int l = 0; l < L; ++l) {
        const int HW = hw[2 * l] * hw[2 * l + 1];
        RAC_CHECK_ARG(ins[l] && outs[l] && HW >= 4 && HW % 4 == 0, "rac_regroup_multi_fwd: level %d (H*W %% 4 must be 0)", l);
        a.lv[l].in = ins[l]; a.lv[l].out = outs[l]; a.lv[l].HW = HW; a.lv[l].tiles_hw = (HW + 63) / 64;
        RAC_CHECK_ARG(total < 2147483647L, "rac_regroup_multi_fwd: grid too large");
        a.lv[l].tile0 = (int)total;
        total += (long)a.lv[l].tiles_hw * a.tiles_c * a.nz;
    }
    RAC_CHECK_ARG(total < 2147483647L, "rac_regroup_multi_fwd: grid too large");
    for (int l = L; l < RG_MAX_LEVELS; ++l)
        a.lv[l] = a.lv[0];
    if (out_dtype == RAC_F32)
        hipLaunchKernelGGL(regroup_multi_kernel<float>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(regroup_multi_kernel<unsigned short>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return rac_launch_status("rac_regroup_multi_fwd");
}

extern "C" int rac_regroup_fwd(const float *in, void *out, int B, int T, int N, int G, int C, int H,
                               int W, int out_dtype, void *stream)
{
    RAC_CHECK_ARG(in && out, "rac_regroup_fwd: null pointer");
    RAC_CHECK_ARG(B >= 0 && T >= 1 && N >= 1 && G >= 1 && C >= 1 && H >= 1 && W >= 1,
                  "rac_regroup_fwd: bad sizes");
    RAC_CHECK_ARG(out_dtype == RAC_F32 || out_dtype == RAC_BF16, "rac_regroup_fwd: dtype %d", out_dtype);
    const long nz = (long)B * T * N * G;
    RAC_CHECK_ARG(nz <= 65535, "rac_regroup_fwd: B*T*N*G=%ld exceeds grid.z", nz);
    if (nz == 0)
        return 0;
    const int HW = H * W;
    dim3 grid((HW + 63) / 64, (C + 63) / 64, (unsigned)nz);
    if (out_dtype == RAC_F32)
        hipLaunchKernelGGL(regroup_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, in, (float *)out, T, N, G, C, HW);
    else
        hipLaunchKernelGGL(regroup_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, in,
                           (unsigned short *)out, T, N, G, C, HW);
    return rac_launch_status("rac_regroup_fwd");
}
