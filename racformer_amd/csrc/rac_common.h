// rac_common.h -- shared helpers for the gfx950 kernels (wave64, HIP only; no torch headers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/racformer_hip.h"

#define RAC_WAVE 64

void rac_set_error(const char *fmt, ...);

// Function attributes (the dynamic-LDS limit) are per device: true the first time kernel slot `id` (0..63, one per kernel that
// raises its limit) is asked for on the CURRENT device, from any thread.  (capi.cpp)
bool rac_first_use_on_device(int id);
enum { RAC_ATTR_GEMM_SPLIT = 0, RAC_ATTR_GENERATOR, RAC_ATTR_CONV3X3, RAC_ATTR_CONV3X3S2, RAC_ATTR_MIXING_F32, RAC_ATTR_MIXING_F16,
       RAC_ATTR_VALUE_PROJ, RAC_ATTR_FPN_CONV };

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

// f32 = hi + lo, both f16 (round-to-nearest): hi carries 11 significant bits, lo the next 11.
struct alignas(8) rac_h4 {
    _Float16 x, y, z, w;
};

__device__ __forceinline__ void rac_split_f16(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
