// rt_device_math.h — the fp32 arithmetic contract of the hot path, device side (gfx950).
//
// Every kernel in this library computes with exactly these operation sequences so that results
// are bit-identical to the CPU oracle: fp32, round-to-nearest-even, no implicit
// contraction (the build passes -ffp-contract=off; every fused multiply-add is an explicit
// __builtin_fmaf), correctly rounded sqrt and division
// (-fhip-fp32-correctly-rounded-divide-sqrt).
//
//   dot(a,b)     = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   length(a)    = sqrt(dot(a,a))          (sqrt_cr below: the same correctly rounded value, fewer instructions)
//   normalize(a) = a * (1 / length(a))
//   cross(a,b).x = fma(a.y,b.z, -(a.z*b.y))   (cyclic)
#pragma once
#include <hip/hip_runtime.h>

namespace rtk {

struct v3 {
    float x, y, z;
};

__device__ __forceinline__ v3 mk(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ v3 scale(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
// a*s + b with one fma per component
__device__ __forceinline__ v3 fma3(v3 a, float s, v3 b) {
    return v3{__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)};
}
__device__ __forceinline__ float dot(v3 a, v3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
// Correctly rounded sqrt.  The compiler's sequence for __builtin_sqrtf is 16 instructions: scale tiny inputs
// by 2^32, v_sqrt_f32, try the two neighbours of the estimate against the fma residuals, unscale, pass zero /
// inf / NaN through.  For 2^-96 <= x < inf - every distance this library ever takes a root of - scaling and
// pass-through do nothing, so those lanes run just the 9-instruction core (the same core, hence the same
// bits); anything else takes the compiler's path.  Exhaustively compared over all 2^32 inputs by
// rt_selftest_math (tests/test_gpu_path_a.py).
__device__ __forceinline__ float sqrt_cr(float x) {
    const uint32_t bits = __float_as_uint(x);
    if (bits - 0x0f800000u < 0x7f800000u - 0x0f800000u) {
        const float s = __builtin_amdgcn_sqrtf(x);  // v_sqrt_f32: within one ulp
        const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
        const float e_below = __builtin_fmaf(-below, s, x), e_above = __builtin_fmaf(-above, s, x);
        float r = e_below <= 0.0f ? below : s;
        r = e_above > 0.0f ? above : r;
        return r;
    }
    return __builtin_sqrtf(x);
}
__device__ __forceinline__ float length(v3 a) { return sqrt_cr(dot(a, a)); }
__device__ __forceinline__ v3 normalize(v3 a) { return scale(a, 1.0f / length(a)); }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
    return v3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
              __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }

// unit-quaternion rotation, shaders/utilities.glsl:26-29 of the reference:
//   t = cross(q.xyz, v) + q.w*v;  return v + 2*cross(q.xyz, t)
__device__ __forceinline__ v3 rotate_q(float qx, float qy, float qz, float qw, v3 v) {
    v3 q = mk(qx, qy, qz);
    v3 c = cross(q, v);
    v3 t = mk(__builtin_fmaf(qw, v.x, c.x), __builtin_fmaf(qw, v.y, c.y), __builtin_fmaf(qw, v.z, c.z));
    v3 c2 = cross(q, t);
    return mk(__builtin_fmaf(2.0f, c2.x, v.x), __builtin_fmaf(2.0f, c2.y, v.y), __builtin_fmaf(2.0f, c2.z, v.z));
}

}  // namespace rtk
