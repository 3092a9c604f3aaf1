// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// This directory restates, in scalar C++ for the host CPU, the arithmetic of the
// rust-raytracer render path so that the GPU library can be checked against it.
// It is never linked into, imported by or executed from the product library; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// "Parity unpinned": the reference ships no tests, golden images or known-answer
// vectors for this path, and its toolchain (cargo/rustc) does not exist in the
// build container, so this restatement cannot be checked against reference output.
// It is pinned only by analytic known-answer pixels and by review against the cited
// reference lines.
//
// Fidelity rules followed throughout (so that results equal what the reference
// computes on x86-64): every quantity is IEEE binary32, expressions are evaluated
// in the reference's written order, no fused multiply-add (build with
// -ffp-contract=off, no fast-math), division is a true division per component,
// sqrt/pow/tan are libm sqrtf/powf/tanf, min/max are fminf/fmaxf.
//
// This file: Point/Vector arithmetic and Color (reference src/vector_math.rs,
// src/color.rs).
#pragma once
#include <cmath>
#include <cstdint>

namespace ref {

// reference src/vector_math.rs:17-22
enum Axis { AX = 0, AY = 1, AZ = 2 };

// One struct serves as both Point (src/vector_math.rs:3-8) and Vector (:10-15);
// the reference's operator overloads (:108-426) are all plain per-component ops.
struct V3 {
    float x, y, z;
    float comp(int axis) const { return axis == AX ? x : (axis == AY ? y : z); }  // :75-81, :93-99
};

static inline V3 add(const V3& a, const V3& b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 sub(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 mul(const V3& a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }       // :220-266
static inline V3 divs(const V3& a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }      // :276-305
static inline V3 neg(const V3& a) { return V3{-a.x, -a.y, -a.z}; }                          // :308-330

// src/vector_math.rs:436-438 -- (x*x + y*y) + z*z
static inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// src/vector_math.rs:428-434
static inline V3 cross(const V3& a, const V3& b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// src/vector_math.rs:25-28
static inline float magnitude(const V3& a) {
    float m2 = a.x * a.x + a.y * a.y + a.z * a.z;
    return sqrtf(m2);
}

// src/vector_math.rs:30-38
static inline V3 normalized(const V3& a) {
    float mag = magnitude(a);
    return divs(a, mag);
}

// src/vector_math.rs:51-53: self - 2.0 * self.dot(normal) * normal
static inline V3 reflected(const V3& v, const V3& n) {
    float k = 2.0f * dot(v, n);
    return sub(v, mul(n, k));
}

// src/vector_math.rs:62-73
static inline V3 refracted(const V3& v, const V3& n, float ri) {
    float dp = dot(v, n);
    float a = 1.0f - ((1.0f - dp * dp) / (ri * ri));
    if (a >= 0.0f) {
        float b = sqrtf(a) + dp / ri;
        return sub(divs(v, ri), mul(n, b));
    }
    return reflected(v, n);  // total internal reflection
}

// src/vector_math.rs:440-448 (normalises its result)
static inline V3 interpolate(const V3& v1, const V3& v2, float s) {
    V3 r{v1.x * s + v2.x * (1.0f - s), v1.y * s + v2.y * (1.0f - s), v1.z * s + v2.z * (1.0f - s)};
    float mag = magnitude(r);
    return divs(r, mag);
}

// src/color.rs:3-38
struct Color {
    float r, g, b;
};
static inline void clamp_upper(Color& c) {  // :11-21, upper clamp only
    if (c.r > 1.0f) c.r = 1.0f;
    if (c.g > 1.0f) c.g = 1.0f;
    if (c.b > 1.0f) c.b = 1.0f;
}
static inline void acc(Color& a, const Color& b) { a.r += b.r; a.g += b.g; a.b += b.b; }  // :88-102
static inline Color scale(const Color& c, float s) { return Color{c.r * s, c.g * s, c.b * s}; }  // :104-126

// ---------------------------------------------------------------------------
// Portable pow.  The reference calls libm powf (src/phong.rs:46,
// src/blinn_phong.rs:40) whose last-bit behaviour is a property of the host's
// glibc build.  To let tests separate "algorithm differs" from "powf rounding
// differs", the oracle can instead use this fixed sequence of IEEE binary64
// +,-,*,/ and integer bit operations.  The GPU library's exact mode carries the
// same sequence, so both produce bit-identical results.  Valid for x > 0 finite;
// other inputs are routed to libm by the caller.
// ---------------------------------------------------------------------------
static inline double pp_from_bits(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
static inline uint64_t pp_to_bits(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }

static inline float portable_powf(float xf, float yf) {
    double x = (double)xf, y = (double)yf;
    // log2(x): x = m * 2^e with m in [sqrt(1/2), sqrt(2))
    uint64_t ux = pp_to_bits(x);
    int64_t e = (int64_t)((ux >> 52) & 0x7ff) - 1023;
    uint64_t mant = ux & 0x000fffffffffffffull;
    double m = pp_from_bits(mant | 0x3ff0000000000000ull);  // [1,2)
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    // ln(m) = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    double lnm = 2.0 * s * p;
    double l2 = (double)e + lnm * 1.4426950408889634;  // log2(x)
    double t = y * l2;
    if (t > 200.0) t = 200.0;    // overflows float anyway
    if (t < -200.0) t = -200.0;  // underflows float anyway
    if (t != t) return (float)t;  // NaN in, NaN out
    // 2^t = 2^k * exp(f ln2), k = round(t), f in [-0.5, 0.5]
    int32_t ki = (int32_t)(t + (t >= 0.0 ? 0.5 : -0.5));
    double kd = (double)ki;
    double f = (t - kd) * 0.6931471805599453;
    double q = 1.0 / 6227020800.0;  // 1/13!
    q = q * f + 1.0 / 479001600.0;
    q = q * f + 1.0 / 39916800.0;
    q = q * f + 1.0 / 3628800.0;
    q = q * f + 1.0 / 362880.0;
    q = q * f + 1.0 / 40320.0;
    q = q * f + 1.0 / 5040.0;
    q = q * f + 1.0 / 720.0;
    q = q * f + 1.0 / 120.0;
    q = q * f + 1.0 / 24.0;
    q = q * f + 1.0 / 6.0;
    q = q * f + 0.5;
    q = q * f + 1.0;
    q = q * f + 1.0;
    double sc = pp_from_bits((uint64_t)(int64_t)(ki + 1023) << 52);
    return (float)(q * sc);
}

}  // namespace ref
