/* detmath.h — deterministic scalar primitives shared by the HIP kernels and the CPU oracle.
 *
 * Why this file exists
 * --------------------
 * Klara.jl draws from Julia's global MT19937 (`randn`, `rand`; e.g. src/samplers/iterate/MALA.jl:84,94,
 * HMC.jl:135,165) and never seeds it, so "identical seeds/inputs" can only mean a build-defined random
 * stream.  The accept mask must be bit-exact between the gfx950 kernels and the CPU oracle; a 1-ulp
 * difference in log/exp/sincos flips an accept when `ratio ≈ log(u)`.  Everything that feeds the accept
 * test is therefore built here from IEEE-754 correctly-rounded operations only (+, -, *, /, sqrt, fma),
 * which gcc on x86-64 (-mfma -ffp-contract=off) and hipcc on gfx950 (-ffp-contract=off) evaluate
 * identically.  Polynomials use explicit fma so the compiler has no freedom.
 *
 * Contents
 *   kd_philox4x32_10   Philox4x32-10 block function (Salmon et al., SC'11).  Bit-compatible with
 *                      rocRAND's rocrand_device::philox4x32_10_engine: key = (seed_lo, seed_hi),
 *                      counter = (block_lo, block_hi, subsequence_lo, subsequence_hi)
 *                      (/opt/rocm/include/rocrand/rocrand_philox4x32_10.h: seed(), discard_*_impl()).
 *   kd_u52             two 32-bit words -> uniform double strictly inside (0,1), exact.
 *   kd_log, kd_exp     FreeBSD-msun-style log/exp (<1 ulp), branch-light, built from + * / fma.
 *   kd_sincos2pi       sin(2*pi*u), cos(2*pi*u) for u in [0,1).
 *   kd_normal_pair     Box-Muller: one Philox block -> two N(0,1) doubles.
 *
 * This header is NOT a restatement of any reference file; the samplers' arithmetic is written
 * separately in oracle/ (following the Julia sources) and in the .hip kernels.
 */
#ifndef KLARA_DETMATH_H
#define KLARA_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KD_FN __host__ __device__ __forceinline__
#else
#define KD_FN static inline __attribute__((always_inline))
#endif

/* ---------------------------------------------------------------- bit casts */
KD_FN uint64_t kd_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
KD_FN double   kd_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
KD_FN double   kd_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* ---------------------------------------------------------------- Philox4x32-10 */
#define KD_PHILOX_M0 0xD2511F53u
#define KD_PHILOX_M1 0xCD9E8D57u
#define KD_PHILOX_W0 0x9E3779B9u
#define KD_PHILOX_W1 0xBB67AE85u

typedef struct { uint32_t x, y, z, w; } kd_u32x4;

KD_FN kd_u32x4 kd_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1)
{
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)KD_PHILOX_M0 * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)KD_PHILOX_M1 * (uint64_t)c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += KD_PHILOX_W0; k1 += KD_PHILOX_W1;
    }
    kd_u32x4 out = { c0, c1, c2, c3 };
    return out;
}

/* Counter layout of the klara stream:
 *   key      = (seed_lo, seed_hi)
 *   c2, c3   = global chain id (lo, hi)           -> rocRAND "subsequence"
 *   c1:c0    = (transition << 24) | slot          -> rocRAND "offset / 4"
 * Transition index -1 (all ones in the 40-bit field) is the initial-state stream.           */
#define KD_SLOT_BITS 24
KD_FN kd_u32x4 kd_stream_block(uint64_t seed, uint64_t chain, uint64_t transition, uint32_t slot)
{
    const uint64_t blk = (transition << KD_SLOT_BITS) | (uint64_t)slot;
    return kd_philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32),
                            (uint32_t)chain, (uint32_t)(chain >> 32),
                            (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* ---------------------------------------------------------------- uniform (0,1) */
/* 52 random bits m -> (m + 0.5) * 2^-52: exact in binary64, never 0 and never 1. */
KD_FN double kd_u52(uint32_t whi, uint32_t wlo)
{
    const double hi = (double)whi;               /* exact */
    const double lo = (double)(wlo >> 12);       /* exact, 20 bits */
    const double m  = kd_fma(hi, 1048576.0, lo); /* hi*2^20 + lo < 2^52, exact */
    return kd_fma(m, 0x1p-52, 0x1p-53);          /* (m + 0.5) * 2^-52, exact (53 bits) */
}

/* kd_log(u) >= kd_log(2^-53) = -36.7368005696771 for every uniform kd_u52 can return, so a Metropolis ratio at
 * or below this guard is rejected whatever the uniform is (lets the kernels skip the draw; same result). */
#define KD_LOG_UMIN_GUARD (-36.74)

/* ---------------------------------------------------------------- log */
/* Algorithm: FreeBSD msun e_log.c reduction x = 2^k * (1+f), sqrt(1/2) <= 1+f < sqrt(2),
 * s = f/(2+f), log(1+f) = f - hfsq + s*(hfsq + R(s^2)); single code path for every f. */
KD_FN double kd_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double x_in = x;
    uint64_t ux = kd_d2u(x);
    int k = 0;
    if ((ux >> 52) == 0) {              /* +subnormal or +0: rescale (zero handled at the end) */
        x *= 0x1p54; ux = kd_d2u(x); k = -54;
    }
    uint32_t hx = (uint32_t)(ux >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    const double m = kd_u2d(((uint64_t)hx << 32) | (ux & 0xffffffffull));
    const double f = m - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * kd_fma(w, kd_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * kd_fma(w, kd_fma(w, kd_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double dk = (double)k;
    double r = kd_fma(s, hfsq + R, dk * ln2_lo) - hfsq + f + dk * ln2_hi;
    /* specials */
    if (x_in == 0.0) r = -__builtin_inf();
    if (x_in < 0.0) r = __builtin_nan("");
    if (!(x_in < __builtin_inf())) r = x_in + x_in;   /* +inf, NaN */
    return r;
}

/* ---------------------------------------------------------------- exp */
/* Algorithm: FreeBSD msun e_exp.c.  x = k*ln2 + r, |r| <= 0.5 ln2, exp(r) = 1 + r*c/(2-c) ... */
KD_FN double kd_exp(double x)
{
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    /* clamp so that (int) conversion is defined; specials fixed at the end */
    double xc = x;
    if (!(xc > -800.0)) xc = -800.0;   /* also catches NaN */
    if (xc > 800.0) xc = 800.0;
    const int k = (int)(invln2 * xc + (xc < 0.0 ? -0.5 : 0.5));
    const double dk = (double)k;
    const double hi = xc - dk * ln2hi;
    const double lo = dk * ln2lo;
    const double r = hi - lo;
    const double xx = r * r;
    const double c = r - xx * kd_fma(xx, kd_fma(xx, kd_fma(xx, kd_fma(xx, P5, P4), P3), P2), P1);
    const double y = 1.0 + (r * c / (2.0 - c) - lo + hi);
    /* scale by 2^k in two exact-power steps (handles subnormal results with one final rounding
     * in the second multiply, like scalbn) */
    const int k1 = k / 2, k2 = k - k1;
    const double s1 = kd_u2d((uint64_t)(0x3ff + k1) << 52);
    const double s2 = kd_u2d((uint64_t)(0x3ff + k2) << 52);
    double res = y * s1 * s2;
    if (x > 709.782712893383973096) res = __builtin_inf();
    if (x < -745.13321910194110842) res = 0.0;
    if (x != x) res = x;
    return res;
}

/* ---------------------------------------------------------------- erf */
/* erf for the tuner's erf_rate_score (src/tuners/AcceptanceRateMCTuner.jl:17).  Evaluated rarely (once per tuning
 * period), so a plain, fixed-trip-count formulation is used instead of the msun rational approximations:
 *   |x| < 3 : erf(x) = 2/sqrt(pi) * exp(-x^2) * sum_{n>=0} x^(2n+1) 2^n / (2n+1)!!      (all terms positive)
 *   |x| >= 3: erfc(x) = exp(-x^2)/sqrt(pi) * 1/(x + (1/2)/(x + 1/(x + (3/2)/(x + ...))))  (60 levels, bottom-up)
 *   |x| >= 6: +-1.                       Accuracy: <= 20 ulp against libm (tests/test_oracle_kats.py) — ample for a step-size score. */
KD_FN double kd_erf(double x)
{
    const double two_over_sqrtpi = 1.12837916709551257390e+00, one_over_sqrtpi = 5.64189583547756286948e-01;
    const double ax = x < 0.0 ? -x : x;
    double r;
    /* exp(-x^2) with the rounding error of x*x folded back in: exp(-(x2 + lo)) = exp(-x2) * (1 - lo) */
    const double x2 = ax * ax;
    const double x2lo = kd_fma(ax, ax, -x2);
    double ex = kd_exp(-x2);
    ex = ex - ex * x2lo;
    if (ax < 3.0) {
        double term = ax, sum = ax;
        for (int n = 0; n < 100; ++n) {
            term = term * (2.0 * x2) / (double)(2 * n + 3);
            sum = sum + term;
        }
        r = two_over_sqrtpi * ex * sum;
    } else if (ax < 6.0) {
        double f = ax;
        for (int k = 60; k >= 1; --k) f = ax + (0.5 * (double)k) / f;
        r = 1.0 - ex * one_over_sqrtpi / f;
    } else {
        r = 1.0;
    }
    if (x != x) return x;
    return x < 0.0 ? -r : r;
}

/* ---------------------------------------------------------------- sin/cos(2*pi*u) */
/* a = 4u in [0,4); q = nearest integer; r = a - q in [-1/2, 1/2] (exact); y = r*pi/2 in [-pi/4, pi/4];
 * msun __kernel_sin/__kernel_cos polynomials on y; quadrant fix-up by q & 3. */
KD_FN void kd_sincos2pi(double u, double* sn, double* cs)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double a = 4.0 * u;
    const int q = (int)(a + 0.5);
    const double r = a - (double)q;
    const double y = kd_fma(r, pio2_hi, r * pio2_lo);
    const double z = y * y;
    const double w = z * z;
    /* sin(y) */
    const double rs = kd_fma(z, kd_fma(z, S4, S3), S2) + z * w * kd_fma(z, S6, S5);
    const double v = z * y;
    const double sy = kd_fma(v, kd_fma(z, rs, S1), y);
    /* cos(y) */
    const double rc = z * kd_fma(z, kd_fma(z, C3, C2), C1) + w * w * kd_fma(z, kd_fma(z, C6, C5), C4);
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double cy = w1 + (((1.0 - w1) - hz) + z * rc);
    const int qq = q & 3;
    const double s_sel = (qq & 1) ? cy : sy;
    const double c_sel = (qq & 1) ? sy : cy;
    *sn = (qq == 2 || qq == 3) ? -s_sel : s_sel;
    *cs = (qq == 1 || qq == 2) ? -c_sel : c_sel;
}

/* ---------------------------------------------------------------- Box-Muller */
/* One Philox block (4 x 32 bit) -> two independent N(0,1):
 *   u1 = u52(x, y), u2 = u52(z, w); rad = sqrt(-2 log u1); z0 = rad cos(2 pi u2); z1 = rad sin(2 pi u2).
 * Element i of a D-vector uses block (i >> 1) of its transition and takes z0 if i is even, z1 if odd. */
KD_FN void kd_normal_pair_ex(kd_u32x4 b, double* z0, double* z1, double* u1_out, double* logu1_out)
{
    const double u1 = kd_u52(b.x, b.y);
    const double u2 = kd_u52(b.z, b.w);
    const double lg = kd_log(u1);
    const double rad = __builtin_sqrt(-2.0 * lg);
    double sn, cs;
    kd_sincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
    *u1_out = u1;            /* = kd_uniform_xy(b): the same words feed the accept uniform of slot ceil(D/2) */
    *logu1_out = lg;
}
KD_FN void kd_normal_pair(kd_u32x4 b, double* z0, double* z1)
{
    double u1, lg;
    kd_normal_pair_ex(b, z0, z1, &u1, &lg);
}

/* uniform for accept tests / slice sampler: words (x,y) of a block (or (z,w) for the second) */
KD_FN double kd_uniform_xy(kd_u32x4 b) { return kd_u52(b.x, b.y); }
KD_FN double kd_uniform_zw(kd_u32x4 b) { return kd_u52(b.z, b.w); }

#endif /* KLARA_DETMATH_H */
