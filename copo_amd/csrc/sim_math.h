// Deterministic device math for the simulator kernels.
//
// The simulator spec is defined on individually rounded IEEE-754 operations: +,-,*,/,sqrt plus the
// polynomials below.  The library is compiled with -ffp-contract=off and HIP's default correctly
// rounded fp32 divide/sqrt, so every value here is reproducible bit for bit on any IEEE machine
// (the CPU oracle re-derives the same spec in scalar C and the parity tests compare raw bits).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace copo {

constexpr float kPi = 3.14159265f;
constexpr float kTwoPi = 6.28318531f;
constexpr float kHalfPi = 1.57079633f;

// Round 6: the hot expressions of the spec are stated with explicit fused multiply-adds (v_fma_f32, one rounding) -- the same
// ones, in the same places, as `fm` (= fmaf) in oracle/copo_oracle.c; -ffp-contract=off keeps the compiler from fusing anything else.
__device__ __forceinline__ float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// sin/cos with 3-term Cody-Waite reduction by pi/2 and the cephes single-precision kernels.
__device__ __forceinline__ void sincos_det(float x, float& s, float& c) {
    const float kf = floorf(fm(x, 0.636619772f, 0.5f));
    const int k = (int)kf;
    float r = fm(-kf, 1.5703125f, x);
    r = fm(-kf, 4.83751297e-4f, r);
    r = fm(-kf, 7.54978996e-8f, r);
    const float z = r * r;
    const float sp = fm(fm(fm(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fm(z, fm(z, fm(fm(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), -0.5f), 1.0f);
    const int q = k & 3;
    const float a = (q & 1) ? cp : sp;
    const float b = (q & 1) ? sp : cp;
    s = (q & 2) ? -a : a;
    c = ((q == 1) || (q == 2)) ? -b : b;
}

__device__ __forceinline__ float atan2_det(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    // one division: tan(a - pi/4) = (mn - mx) / (mn + mx) above tan(pi/8), mn / mx below
    const bool hi = mn > 0.414213562f * mx;
    const float t = (hi ? mn - mx : mn) / (hi ? mn + mx : mx);
    const float off = hi ? 0.785398163f : 0.0f;
    const float z = t * t;
    const float p = fm(fm(fm(fm(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f) * z, t, t);
    float r = off + p;
    if (ay > ax) r = kHalfPi - r;
    if (x < 0.0f) r = kPi - r;
    return y < 0.0f ? -r : r;
}

__device__ __forceinline__ float log_det(float u) {  // u in (0, 1], normal
    uint32_t b = __float_as_uint(u);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float p = (((0.111111111f * z + 0.142857143f) * z + 0.2f) * z + 0.333333333f) * z + 1.0f;
    return (float)e * 0.693147181f + 2.0f * s * p;
}

__device__ __forceinline__ float wrap_pi(float a) {
    if (a > kPi) a -= kTwoPi;
    if (a < -kPi) a += kTwoPi;
    return a;
}

__device__ __forceinline__ float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// counter-based RNG: one 32-bit draw per (env seed, a, b, c, stream)
__device__ __forceinline__ uint32_t hash_rng(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t stream) {
    uint32_t h = mix32((uint32_t)seed + 0x9E3779B9u);
    h = mix32(h ^ (uint32_t)(seed >> 32));
    h = mix32(h ^ a);
    h = mix32(h ^ b);
    h = mix32(h ^ c);
    h = mix32(h ^ stream);
    return h;
}

__device__ __forceinline__ float uniform01(uint32_t h) { return ((float)(h >> 9) + 0.5f) * 1.1920929e-7f; }

}  // namespace copo
