// Device-side float64 vector helpers, Go-compatible min/max and the Philox stream.
// The arithmetic order of the reference's r3 package is kept and the build disables
// floating point contraction, so float64 results match the reference algorithm's.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#define RT_DEV __device__ __forceinline__
// a pointer read from memory (not a kernel argument) is generic to the compiler; one that is known to point into device memory is cast to this
#define RT_GLOBAL_AS __attribute__((address_space(1)))
#define RT_HD __host__ __device__ __forceinline__

namespace rt {

constexpr double kEps = 1e-6;  // phys/math_extra.go:6
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kMaxFloat = 1.7976931348623157e308;  // math.MaxFloat64

struct V3 {
    double x, y, z;
};

RT_HD V3 mk(double x, double y, double z) { return V3{x, y, z}; }
RT_HD V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
RT_HD void st3(double* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
RT_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }   // r3/vec.go:16
RT_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }   // r3/vec.go:21
RT_HD V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }   // r3/vec.go:26
RT_HD V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }     // r3/vec.go:36
RT_HD V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }     // r3/vec.go:41
RT_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     // r3/vec.go:48
RT_HD V3 cross(V3 a, V3 b) {                                                   // r3/vec.go:54
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RT_HD double length(V3 a) { return sqrt(dot(a, a)); }                          // r3/vec.go:85
RT_HD V3 unit(V3 a) {                                                          // r3/vec.go:91-97
    double l = length(a);
    if (l == 0) return {0, 0, 0};
    return a / l;
}
RT_HD double comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// Go's math.Max / math.Min: +Inf / -Inf win, then NaN propagates, then signed zeros.
RT_HD double go_max(double x, double y) {
    if (x == (double)INFINITY || y == (double)INFINITY) return (double)INFINITY;
    if (x != x || y != y) return (double)NAN;
    if (x == 0 && x == y) return signbit(x) ? y : x;
    return x > y ? x : y;
}
RT_HD double go_min(double x, double y) {
    if (x == -(double)INFINITY || y == -(double)INFINITY) return -(double)INFINITY;
    if (x != x || y != y) return (double)NAN;
    if (x == 0 && x == y) return signbit(x) ? x : y;
    return x < y ? x : y;
}

RT_HD bool bad3(V3 a) {  // IsNaN || IsInf, r3/vec.go:106-113
    return !(isfinite(a.x) && isfinite(a.y) && isfinite(a.z));
}

// ---- random streams (stand in for phys/rand.go's math/rand wrapper) -------------------------
// One Philox4x32-10 stream per ray, keyed by the seed, counted by (pixel, sample, path, draw/2).

// n / d in three instructions when r = 1 / d is at hand: q0 = RN(n r) is within an ulp of the quotient, the remainder
// n - q0 d is exact in a fused multiply-add, and RN(q0 + rem r) is the correctly rounded quotient (Markstein's theorem;
// r must be the correctly rounded reciprocal, everything in the normal range).  The result is bit for bit the IEEE
// division the reference performs.
RT_HD double div_by(double n, double d, double r) {
    const double q0 = n * r;
    const double rem = __builtin_fma(-q0, d, n);
    return __builtin_fma(rem, r, q0);
}

RT_HD void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

struct Rng {
    uint32_t k0, k1, pixel, sample, path, n;
    // A Philox block is 128 bits = two draws: draw n is half n & 1 of the block with counter word n >> 1.  The block of
    // the last draw is kept, so two consecutive draws cost one block (the stream is the same either way).
    uint32_t have = 0xffffffffu;  // counter word of the block in `blk`
    uint32_t blk[4];
    RT_HD double float64() {  // uniform in [0,1), 53 bits, like math/rand's Float64
        const uint32_t word = n >> 1;
        if (word != have) {
            blk[0] = pixel; blk[1] = sample; blk[2] = path; blk[3] = word;
            philox4x32_10(blk, k0, k1);
            have = word;
        }
        const uint64_t bits = (n & 1) ? ((uint64_t)blk[3] << 32 | blk[2]) : ((uint64_t)blk[1] << 32 | blk[0]);
        n++;
        return (double)(bits >> 11) * 0x1.0p-53;
    }
};

RT_HD V3 in_unit_sphere(Rng& r) {  // rand.go:26-39
    for (;;) {
        double a = r.float64(), b = r.float64(), c = r.float64();
        V3 p = mk(a, b, c) * 2.0 - mk(1, 1, 1);
        if (length(p) < 1.0) return p;
    }
}
RT_HD V3 in_unit_disk(Rng& r) {  // rand.go:62-75
    for (;;) {
        double a = r.float64(), b = r.float64();
        V3 p = mk(a, b, 0) * 2.0 - mk(1, 1, 0);
        if (dot(p, p) < 1.0) return p;
    }
}
RT_HD V3 cosine_weighted_hemisphere(Rng& r, V3 n) {  // rand.go:79-105
    double u1 = r.float64();
    double u2 = r.float64();
    double r1 = sqrt(u1);
    double theta = 2 * kPi * u2;
    double x = r1 * cos(theta);
    double y = r1 * sin(theta);
    double z = sqrt(1 - u1);
    V3 tangent = fabs(n.x) > fabs(n.y) ? unit(mk(-n.z, 0, n.x)) : unit(mk(0, n.z, -n.y));
    V3 bitangent = cross(n, tangent);
    return unit(tangent * x + bitangent * y + n * z);
}

}  // namespace rt
