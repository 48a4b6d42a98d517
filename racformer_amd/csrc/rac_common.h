// rac_common.h -- shared helpers for the gfx950 kernels (wave64, HIP only; no torch headers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/racformer_hip.h"

#define RAC_WAVE 64

void rac_set_error(const char *fmt, ...);

// Function attributes (the dynamic-LDS limit) are per device: sets it for kernel slot `id` (0..63, one per kernel that raises its
// limit) on the CURRENT device unless an earlier call already did, from any thread; 0 or the HIP error (rac_last_error set).  (capi.cpp)
int rac_set_dynamic_lds_once(int id, const void *func, int bytes);
enum { RAC_ATTR_GEMM_SPLIT = 0, RAC_ATTR_GENERATOR, RAC_ATTR_CONV3X3, RAC_ATTR_CONV3X3S2, RAC_ATTR_MIXING_F32, RAC_ATTR_MIXING_F16,
       RAC_ATTR_VALUE_PROJ, RAC_ATTR_FPN_CONV, RAC_ATTR_GENERATOR4, RAC_ATTR_CONV3X3_Q16, RAC_ATTR_VALUE_PROJ_Q16, RAC_ATTR_MIXING_SAMPLED,
       RAC_ATTR_CD_GRU, RAC_ATTR_CD_S2_8, RAC_ATTR_CD_S2_2, RAC_ATTR_CD_S2_3, RAC_ATTR_CD_IMG_2, RAC_ATTR_CD_F32_1, RAC_ATTR_CD_F32_2,
       RAC_ATTR_CD_F32_4, RAC_ATTR_VALUE_PROJ_BIAS, RAC_ATTR_VALUE_PROJ_Q16_BIAS };

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (straight-line code with the index usable in constexpr contexts)
template <int I>
struct rac_ic {
    static constexpr int value = I;
};
template <int B, int E, typename F>
__device__ __forceinline__ void rac_static_for(F &&f)
{
    if constexpr (B < E) {
        f(rac_ic<B>{});
        rac_static_for<B + 1, E>(f);
    }
}

#define RAC_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            rac_set_error(__VA_ARGS__);     \
            return RAC_E_ARG;               \
        }                                   \
    } while (0)

static inline int rac_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rac_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// bf16 <-> f32 (bit-level; bf16 is the top half of an IEEE f32)
__device__ __forceinline__ float rac_bf16_to_f32(unsigned short v)
{
    return __uint_as_float(((unsigned)v) << 16);
}

struct alignas(16) rac_f4 {
    float x, y, z, w;
};

__device__ __forceinline__ rac_f4 rac_ld4(const float *p)
{
    return *reinterpret_cast<const rac_f4 *>(p);
}
// Streaming forms for data a kernel touches exactly once (the pyramid transpose, the mixing kernel's operands and output): non-temporal,
// so that the lines do not displace what the gather kernels of the samples running beside this one keep re-reading from L2
// (several plans in flight, racformer_amd/graph.py).  RAC_STREAM_NT = 0: plain accesses (A/B builds).
#ifndef RAC_STREAM_NT
#define RAC_STREAM_NT 1
#endif
typedef float rac_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rac_f4 rac_ld4_stream(const float *p)
{
#if RAC_STREAM_NT
    const rac_f4v v = __builtin_nontemporal_load(reinterpret_cast<const rac_f4v *>(p));
    return rac_f4{v.x, v.y, v.z, v.w};
#else
    return rac_ld4(p);
#endif
}
__device__ __forceinline__ void rac_st4_stream(float *p, const rac_f4 v)
{
#if RAC_STREAM_NT
    __builtin_nontemporal_store((rac_f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<rac_f4v *>(p));
#else
    *reinterpret_cast<rac_f4 *>(p) = v;
#endif
}
// 4 bf16 (8 bytes) -> 4 floats
__device__ __forceinline__ rac_f4 rac_ld4(const unsigned short *p)
{
    uint2 r = *reinterpret_cast<const uint2 *>(p);
    rac_f4 o;
    o.x = __uint_as_float(r.x << 16);
    o.y = __uint_as_float(r.x & 0xffff0000u);
    o.z = __uint_as_float(r.y << 16);
    o.w = __uint_as_float(r.y & 0xffff0000u);
    return o;
}

// Four channels of one bilinear tap into a lane's running sums (the gather kernels' inner operation).
// Scalar v_fma_f32, deliberately NOT v_pk_fma_f32 (two channels per instruction, what round 2 used): beside another stream's
// MFMA kernels (mixing_c64_f16x3_kernel, conv3x3_f16x3_kernel -- several samples in flight, racformer_amd/graph.py) the packed
// form produced wrong sums in 4-lane granules in every gather kernel while the scalar form never did (round 3; tools/race_victims.py is the tool that remains:
// rac_msmv_fwd beside a looping mixing kernel, 110 of 120 launches deviating with v_pk_fma_f32, 0 of 120 with v_fma_f32; same
// loads, same waits, plain global loads instead of buffer loads made no difference).  Same rounding either way (IEEE fma per
// component), so results are bit-identical to round 2's.
// The packed form exists for diagnostic builds only (tools/build_variant.sh defines RAC_DIAGNOSTIC_BUILD; tools/race_victims.py
// runs them): a product build that asks for it does not compile.
#ifndef RAC_GATHER_PACKED_FMA
#define RAC_GATHER_PACKED_FMA 0
#endif
#if RAC_GATHER_PACKED_FMA && !defined(RAC_DIAGNOSTIC_BUILD)
#error "RAC_GATHER_PACKED_FMA is a diagnostic switch (DESIGN 3.12): packed FP32 accumulation goes wrong beside another stream's MFMA kernels"
#endif
typedef float rac_f2v __attribute__((ext_vector_type(2)));
struct rac_acc4 {
#if RAC_GATHER_PACKED_FMA
    rac_f2v a01, a23;
#else
    float a0, a1, a2, a3;
#endif
};
__device__ __forceinline__ rac_acc4 rac_acc4_zero()
{
#if RAC_GATHER_PACKED_FMA
    return rac_acc4{(rac_f2v){0.f, 0.f}, (rac_f2v){0.f, 0.f}};
#else
    return rac_acc4{0.f, 0.f, 0.f, 0.f};
#endif
}
__device__ __forceinline__ void rac_tap_fma(rac_acc4 &acc, float x, float y, float z, float w4, float w)
{
#if RAC_GATHER_PACKED_FMA
    const rac_f2v w2 = {w, w};
    acc.a01 = __builtin_elementwise_fma((rac_f2v){x, y}, w2, acc.a01);
    acc.a23 = __builtin_elementwise_fma((rac_f2v){z, w4}, w2, acc.a23);
#else
    acc.a0 = __builtin_fmaf(x, w, acc.a0);
    acc.a1 = __builtin_fmaf(y, w, acc.a1);
    acc.a2 = __builtin_fmaf(z, w, acc.a2);
    acc.a3 = __builtin_fmaf(w4, w, acc.a3);
    asm volatile("" : "+v"(acc.a0), "+v"(acc.a1), "+v"(acc.a2), "+v"(acc.a3));   // (keeps the SLP vectoriser from re-packing them)
#endif
}
__device__ __forceinline__ void rac_acc4_get(const rac_acc4 &acc, float &x, float &y, float &z, float &w)
{
#if RAC_GATHER_PACKED_FMA
    x = acc.a01.x; y = acc.a01.y; z = acc.a23.x; w = acc.a23.y;
#else
    x = acc.a0; y = acc.a1; z = acc.a2; w = acc.a3;
#endif
}

// f32 = hi + lo, both f16 (round-to-nearest): hi carries 11 significant bits, lo the next 11.
struct alignas(8) rac_h4 {
    _Float16 x, y, z, w;
};

__device__ __forceinline__ void rac_split_f16(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// The activations' power-of-two scale of the split-precision convolution kernels (conv3x3.hip, conv_direct.hip): brings amax -- the
// maximum |value| of an activation image, or a bound of it -- into [2^13, 2^14) (1 for amax == 0 or non-finite input), so that hi is a
// normal f16 for every value and lo keeps the next 11 bits down to values 2^-13 of the maximum.
__device__ __forceinline__ float rac_act_scale(float amax)
{
    if (!(amax > 1.0e-30f) || !(amax < 3.0e38f))
        return 1.f;
    const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 126;   // amax = m * 2^ex, m in [0.5, 1)
    return __uint_as_float((unsigned)(14 - ex + 127) << 23);             // 2^(14 - ex)
}

// int16 block storage of a BEV value stream (quant.hip; the q16 epilogues of conv3x3.hip and value_proj.hip): value = q * dn with
// dn = 2^(e - 14), e = floor(log2(max |block|)).  Exponent arithmetic only: eb = biased exponent of the block maximum, clamped so that
// both powers of two are normal floats (an all-zero / denormal block gets q = 0 with the smallest scale).  A block whose maximum is
// inf or NaN gets a NaN scale: every value of the block reads back as NaN, as it would propagate through the fp32 stream (a
// diverged model must not look healthy because its storage saturates).  The block maximum is taken on the BIT PATTERNS
// of |x| as unsigned integers (rac_absbits / max): monotonic for finite values, and a NaN (pattern above inf's) wins, where fmaxf
// would drop it.
__device__ __forceinline__ unsigned rac_absbits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
__device__ __forceinline__ unsigned rac_absbits4(float x, float y, float z, float w)
{
    return max(max(rac_absbits(x), rac_absbits(y)), max(rac_absbits(z), rac_absbits(w)));
}
__device__ __forceinline__ void rac_q16_factors(unsigned block_max_bits, float &up, float &dn)
{
    int eb = (int)((block_max_bits >> 23) & 255u);
    if (eb == 255) {
        up = 0.f;
        dn = __uint_as_float(0x7fc00000u);
        return;
    }
    eb = eb < 15 ? 15 : eb;
    up = __uint_as_float((unsigned)(268 - eb) << 23);      // 2^(14 - (eb - 127)): |x| * up < 2^15
    dn = __uint_as_float((unsigned)(eb - 14) << 23);       // 2^((eb - 127) - 14)
}
__device__ __forceinline__ int rac_q16(float x, float up)
{
    return (int)fminf(fmaxf(rintf(x * up), -32767.f), 32767.f);
}
// four values -> four int16 (8 bytes)
__device__ __forceinline__ uint2 rac_q16x4(float x, float y, float z, float w, float up)
{
    uint2 o;
    o.x = ((unsigned)rac_q16(x, up) & 0xffffu) | ((unsigned)rac_q16(y, up) << 16);
    o.y = ((unsigned)rac_q16(z, up) & 0xffffu) | ((unsigned)rac_q16(w, up) << 16);
    return o;
}

// Diagnostic builds only (-DRAC_CLOCK_STAMPS through tools/build_variant.sh; tools/kernel_clock.py): the clock a matrix-core kernel
// holds under its own load.  Wave 0 of every workgroup stamps s_memtime (shader clock ticks) and s_memrealtime (100 MHz) once
// around its main loop; in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, 'DVFS give-back' item 6).
// The stamps go to a device array of their own that nothing else reads; not part of the product library or its ABI.
#if defined(RAC_CLOCK_STAMPS) && defined(RAC_DIAGNOSTIC_BUILD)
#define RAC_CLOCK_WGS 1024
#define RAC_CLOCK_DECL(name) __device__ unsigned long long name##_clock_buf[RAC_CLOCK_WGS * 4];
#define RAC_CLOCK_BEGIN() const unsigned long long clk_t0_ = __builtin_amdgcn_s_memtime(), clk_r0_ = __builtin_amdgcn_s_memrealtime()
#define RAC_CLOCK_END(name, wg)                                                  \
    do {                                                                         \
        if ((threadIdx.x & 63) == 0 && threadIdx.x < 64 && (wg) < RAC_CLOCK_WGS) { \
            name##_clock_buf[(wg) * 4 + 0] = clk_t0_;                            \
            name##_clock_buf[(wg) * 4 + 1] = clk_r0_;                            \
            name##_clock_buf[(wg) * 4 + 2] = __builtin_amdgcn_s_memtime();       \
            name##_clock_buf[(wg) * 4 + 3] = __builtin_amdgcn_s_memrealtime();   \
        }                                                                        \
    } while (0)
#define RAC_CLOCK_READER(name)                                                                                           \
    extern "C" int rac_dbg_clock_##name(unsigned long long *host_out, int n_wgs)                                         \
    {                                                                                                                    \
        if (n_wgs > RAC_CLOCK_WGS)                                                                                       \
            n_wgs = RAC_CLOCK_WGS;                                                                                       \
        hipDeviceSynchronize();                                                                                          \
        return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(name##_clock_buf), sizeof(unsigned long long) * 4 * n_wgs, 0, \
                                        hipMemcpyDeviceToHost);                                                          \
    }
#else
#define RAC_CLOCK_DECL(name)
#define RAC_CLOCK_BEGIN()
#define RAC_CLOCK_END(name, wg)
#define RAC_CLOCK_READER(name)
#endif
