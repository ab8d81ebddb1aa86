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
 *   kd_log             FreeBSD-msun-style log (<1 ulp), every special case; kd_exp: 128-entry table, division-free (<1 ulp).
 *   kd_log_u01         table-driven, division-free log for the uniforms (radius and Metropolis tests).
 *   kd_sincos2pi       sin(2*pi*u), cos(2*pi*u) for u in [0,1): 256-entry table + rotation.
 *   kd_normal_pair_w   Box-Muller on 64 bits: half a Philox block -> two N(0,1) doubles (44-bit radius uniform, 20-bit angle at cell centres);
 *                      kd_normal_pair_at: the proposal normals of element pair p — one block serves pairs p and p + 8.
 *
 * This header is NOT a restatement of any reference file; the samplers' arithmetic is written
 * separately in oracle/ (following the Julia sources) and in the .hip kernels.
 */
#ifndef KLARA_DETMATH_H
#define KLARA_DETMATH_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

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

/* a ^ b ^ c: one v_bitop3_b32 (truth table 0x96) on gfx950 instead of two v_xor_b32 — 20 fewer VALU issues per block */
#if defined(__HIP_DEVICE_COMPILE__)
#define KD_XOR3(a, b, c) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96))
#else
#define KD_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#endif

KD_FN kd_u32x4 kd_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1)
{
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)KD_PHILOX_M0 * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)KD_PHILOX_M1 * (uint64_t)c2;
        const uint32_t n0 = KD_XOR3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = KD_XOR3((uint32_t)(p0 >> 32), c3, k1);
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
/* 52 random bits m = whi:wlo>>12 -> (m + 0.5) * 2^-52: exact in binary64, never 0 and never 1.
 * Built without int->double conversions: uu = 1 + m 2^-52 is assembled in [1,2) from the bits, and
 * uu - (1 - 2^-53) = (2m + 1) 2^-53 is representable, so the subtraction is exact. */
KD_FN uint64_t kd_unit_bits(uint32_t whi, uint32_t wlo)      /* bits of 1 + m 2^-52 */
{
    const uint32_t uh = 0x3ff00000u | (whi >> 12);
    const uint32_t ul = (whi << 20) | (wlo >> 12);
    return ((uint64_t)uh << 32) | (uint64_t)ul;
}
KD_FN double kd_u52(uint32_t whi, uint32_t wlo)
{
    return kd_u2d(kd_unit_bits(whi, wlo)) - 0x1.fffffffffffffp-1;
}

/* 44 random bits m = wa:wb[11:0] -> (m + 0.5) * 2^-44, built the same way (uu - (1 - 2^-45) = (2m + 1) 2^-45: exact).  The radius
 * uniform of the proposal normals and the accept uniform of the Metropolis tests: half a Philox block each (kd_normal_pair_w). */
KD_FN uint64_t kd_unit_bits44(uint32_t wa, uint32_t wb)      /* bits of 1 + m 2^-44 */
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t uh = __builtin_amdgcn_alignbit(0x3ffu, wa, 12);          /* 0x3ff00000 | wa >> 12 */
    const uint32_t ul = __builtin_amdgcn_alignbit(wa, wb << 20, 12);        /* wa << 20 | (wb & 0xfff) << 8 */
#else
    const uint32_t uh = 0x3ff00000u | (wa >> 12);
    const uint32_t ul = (wa << 20) | ((wb << 20) >> 12);
#endif
    return ((uint64_t)uh << 32) | (uint64_t)ul;
}
KD_FN double kd_u44(uint32_t wa, uint32_t wb)
{
    return kd_u2d(kd_unit_bits44(wa, wb)) - 0x1.fffffffffffp-1;
}

/* kd_log_u01(u) >= kd_log_u01(2^-45) = -31.1916... for every uniform kd_u44 can return, so a Metropolis ratio at
 * or below this guard is rejected whatever the accept uniform is (lets the kernels skip the draw; same result). */
#define KD_LOG_UMIN_GUARD (-31.2)

/* ---------------------------------------------------------------- log */
/* Algorithm: FreeBSD msun e_log.c reduction x = 2^k * (1+f), sqrt(1/2) <= 1+f < sqrt(2),
 * s = f/(2+f), log(1+f) = f - hfsq + s*(hfsq + R(s^2)); single code path for every f.
 * The reduction, the polynomial coefficients Lg1..Lg7 and the ln2 split are those of e_log.c, whose notice reads:
 * ====================================================
 * Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 * Developed at SunSoft, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this
 * software is freely granted, provided that this notice
 * is preserved.
 * ==================================================== */
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

/* ---------------------------------------------------------------- log of a uniform */
/* kd_log_u01(x) for positive normal finite x (every kd_u52 value): the logarithm the Box-Muller radius and the
 * Metropolis tests use.  Division-free table method (the scheme of Arm's optimized-routines log, restated):
 *   x = 2^k z, z in [0.6875, 1.375); bin i = leading 7 mantissa bits of z relative to 0.6875;
 *   r = z*invc_i - 1 (one fma, |r| <= 2^-7);  log x = (k ln2_hi + logc_i) + r + [k ln2_lo - r^2/2 + r^3 q(r)]
 * with q the degree-5 Taylor tail of log1p.  The two bins touching 1 have c = 1, logc = 0, so r = x - 1 exactly and
 * the result keeps its relative accuracy as x -> 1.  < 1 ulp (tests/test_oracle_kats.py).  No special cases:
 * zero, subnormal, negative, inf and NaN inputs are outside its contract (kd_log handles those). */
#include "detmath_tables.h"
static const double kd_logtab_host[256] __attribute__((aligned(16))) = KD_LOGTAB_INIT;
static const double kd_sctab_host[512] __attribute__((aligned(16))) = KD_SCTAB_INIT;
static const double kd_exptab_host[256] __attribute__((aligned(16))) = KD_EXPTAB_INIT;
static const double kd_l12tab_host[256] __attribute__((aligned(16))) = KD_L12TAB_INIT;
#if defined(__HIPCC__)
/* On the GPU the tables live in LDS (8 KB per workgroup): the lookups are per-lane gathers, and as DS reads they are
 * tracked by lgkmcnt, so waiting for one never drains the HBM prefetch that is in flight on vmcnt.  Every kernel that
 * draws normals, takes a uniform's log or calls kd_exp calls kd_tables_to_lds() once, first thing. */
static __device__ const double kd_logtab_dev[256] __attribute__((aligned(16))) = KD_LOGTAB_INIT;
static __device__ const double kd_sctab_dev[512] __attribute__((aligned(16))) = KD_SCTAB_INIT;
static __device__ const double kd_exptab_dev[256] __attribute__((aligned(16))) = KD_EXPTAB_INIT;
/* kd_log12's table is NOT part of the 8 KB block every kernel stages (a launch gets 64 KB of LDS without asking, and the staged-closure
 * layouts are sized against what the block leaves): the generic function reads it from memory (2 KB, cache resident), the logistic
 * kernels keep their own copy in LDS behind the data rows and gather from there (klara_kernels.h LogisticTarget). */
static __device__ const double kd_l12tab_dev[256] __attribute__((aligned(16))) = KD_L12TAB_INIT;
__shared__ double kd_tab_lds[1024] __attribute__((aligned(16)));
__device__ __forceinline__ void kd_tables_to_lds()
{
    for (int i = (int)threadIdx.x; i < 1024; i += (int)blockDim.x)
        kd_tab_lds[i] = i < 256 ? kd_logtab_dev[i] : (i < 768 ? kd_sctab_dev[i - 256] : kd_exptab_dev[i - 768]);
    __syncthreads();
}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define KD_LOGTAB(i) kd_tab_lds[i]
#define KD_SCTAB(i) kd_tab_lds[256 + (i)]
#define KD_EXPTAB(i) kd_tab_lds[768 + (i)]
#define KD_L12TAB(i) kd_l12tab_dev[i]
#else
#define KD_LOGTAB(i) kd_logtab_host[i]
#define KD_SCTAB(i) kd_sctab_host[i]
#define KD_EXPTAB(i) kd_exptab_host[i]
#define KD_L12TAB(i) kd_l12tab_host[i]
#endif

/* The evaluation is split at its table read — reduce (bits -> bin, exponent, reduced argument), the gather, finish — so that a caller
 * with several independent arguments (the data rows of the logistic targets, klara_kernels.h) can issue all the gathers of a batch before the
 * first finish; kd_log_u01 itself is the three in sequence, so every caller computes the same bits. */
KD_FN void kd_log_u01_reduce(double x, uint32_t* i, int* k, double* z)
{
    const uint64_t ux = kd_d2u(x);
    const uint32_t hx = (uint32_t)(ux >> 32);
    const uint32_t tmp = hx - 0x3fe60000u;
    *i = (tmp >> 13) & 127u;
    *k = (int32_t)tmp >> 20;                              /* arithmetic shift: floor */
    const uint32_t hz = hx - (tmp & 0xfff00000u);
    *z = kd_u2d(((uint64_t)hz << 32) | (ux & 0xffffffffull));
}
KD_FN double kd_log_u01_finish(double z, int k, double invc, double logc)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;  /* ln2_hi: 32 trailing zero bits */
    const double B0 = 0x1.5555555555555p-2, B1 = -0.25, B2 = 0x1.999999999999ap-3, B3 = -0x1.5555555555555p-3,
                 B4 = 0x1.2492492492492p-3, B5 = -0.125;
    const double r = kd_fma(z, invc, -1.0);
    const double dk = (double)k;
    const double w = kd_fma(dk, ln2_hi, logc);            /* dk*ln2_hi exact */
    const double hi = w + r;
    const double lo = kd_fma(dk, ln2_lo, (w - hi) + r);   /* Fast2Sum: |w| >= |r| or w == 0 */
    const double r2 = r * r;
    const double q = kd_fma(r2, kd_fma(r2, kd_fma(r, B5, B4), kd_fma(r, B3, B2)), kd_fma(r, B1, B0));
    return hi + kd_fma(r * r2, q, kd_fma(r2, -0.5, lo));
}
KD_FN double kd_log_u01(double x)
{
    uint32_t i; int k; double z;
    kd_log_u01_reduce(x, &i, &k, &z);
    return kd_log_u01_finish(z, k, KD_LOGTAB(2 * i), KD_LOGTAB(2 * i + 1));
}

/* ---------------------------------------------------------------- exp */
/* Division-free table method (the scheme of Arm's optimized-routines exp, restated): x = k ln2/128 + r, |r| <= ln2/256;
 * exp(x) = 2^(k>>7) * T[k & 127] * (1 + p(r)), T = 2^(i/128) as a (hi, lo) pair, p = r + r^2/2 + ... + r^5/120.
 * < 1 ulp (tests/test_oracle_kats.py).  Overflow -> +inf, underflow -> subnormals / 0 through a two-factor power-of-two
 * scaling (one rounding), NaN -> NaN. */
KD_FN double kd_exp(double x)
{
    const double C2 = 0.5, C3 = 0x1.5555555555555p-3, C4 = 0x1.5555555555555p-5, C5 = 0x1.1111111111111p-7;
    /* clamp so that (int) conversion is defined; specials fixed at the end */
    double xc = x;
    if (!(xc > -800.0)) xc = -800.0;   /* also catches NaN */
    if (xc > 800.0) xc = 800.0;
    const int k = (int)(KD_INVLN2N * xc + (xc < 0.0 ? -0.5 : 0.5));
    const double dk = (double)k;
    const double r = kd_fma(dk, -KD_LN2N_LO, kd_fma(dk, -KD_LN2N_HI, xc));     /* dk*LN2N_HI is exact (32-bit constant) */
    const int idx = k & 127, e = k >> 7;                                      /* arithmetic shift: floor */
    const double th = KD_EXPTAB(2 * idx), tl = KD_EXPTAB(2 * idx + 1);
    const double r2 = r * r;
    const double p = kd_fma(r2 * r2, kd_fma(r, C5, C4), kd_fma(r2, kd_fma(r, C3, C2), r));
    const double y = th + kd_fma(th, p, tl);
    /* scale by 2^e in two exact-power steps (handles subnormal results with one final rounding
     * in the second multiply, like scalbn) */
    const int e1 = e / 2, e2 = e - e1;
    const double s1 = kd_u2d((uint64_t)(0x3ff + e1) << 52);
    const double s2 = kd_u2d((uint64_t)(0x3ff + e2) << 52);
    double res = y * s1 * s2;
    if (x > 709.782712893383973096) res = __builtin_inf();
    if (x < -745.13321910194110842) res = 0.0;
    if (x != x) res = x;
    return res;
}

/* exp(-a) for a >= 0, the form the logistic rows need (t = exp(-|Xp|) in (0, 1]): kd_exp's table and polynomial with the reduction of
 * Arm's optimized-routines exp — k = round-to-nearest(x 128/ln2) read from the low mantissa bits of fma(x, 128/ln2, 1.5 * 2^52), and
 * k as a double by one exact subtraction: one fma, one sub and no int <-> double conversion where kd_exp takes a mul, an add and two
 * conversions — without what the range does not need: no overflow side, no NaN or subnormal handling, and the power of two is added to
 * the exponent field (results are normal numbers, so this equals kd_exp's two exact multiplications).  Beyond a = 708, where the value
 * would be subnormal, the argument is clamped: the result is exp(-708) = 3.3e-308 instead of a number below 2.3e-308 — an absolute error
 * below 3.3e-308 (the only callers form 1 + t, where it vanishes, and t / (1 + t)).  A NaN argument gives exp(-708) as well (fmax
 * returns the other operand): kd_softplus_logistic adds the pass-through; the batched data rows of the logistic kernels do NOT (see
 * kd_softplus_logistic_rows below for why that changes no result).  < 1 ulp up to 708 (tests/test_oracle_kats.py); NOT bit-identical
 * with kd_exp(-a): k is rounded to nearest-even here and towards zero-after-offset there, so a few arguments near the ties between two
 * table entries are reduced to the other neighbour.
 * Split at its table read — reduce, the gather (KD_EXPTAB(2 (k & 127)), (2 (k & 127) + 1)), polynomial, combine — so that a caller with
 * several independent arguments (the data rows of the logistic targets, klara_kernels.h) can have all the gathers of a batch in flight
 * before the first is consumed; kd_exp_neg itself is the four in sequence. */
#define KD_EXP_SHIFT 0x1.8p52
KD_FN void kd_exp_neg_reduce(double a, int* k, double* r)
{
    const double xc = __builtin_fmax(-a, -708.0);
    const double kd = kd_fma(xc, KD_INVLN2N, KD_EXP_SHIFT);            /* in [2^52, 2^53): ulp 1, so the sum is rounded to an integer */
    *k = (int)(uint32_t)kd_d2u(kd);                                     /* ... whose two's complement sits in the low mantissa bits */
    const double dk = kd - KD_EXP_SHIFT;                                /* exact */
    *r = kd_fma(dk, -KD_LN2N_LO, kd_fma(dk, -KD_LN2N_HI, xc));
}
KD_FN double kd_exp_neg_poly(double r)                 /* p(r) = r + r^2/2 + ... + r^5/120: needs no table value */
{
    const double C2 = 0.5, C3 = 0x1.5555555555555p-3, C4 = 0x1.5555555555555p-5, C5 = 0x1.1111111111111p-7;
    const double r2 = r * r;
    return kd_fma(r2 * r2, kd_fma(r, C5, C4), kd_fma(r2, kd_fma(r, C3, C2), r));
}
KD_FN double kd_exp_neg_combine(int k, double p, double th, double tl)
{
    const double y = th + kd_fma(th, p, tl);
    /* 2^(k >> 7) into the exponent field: only the high word changes ((k >> 7) << 20 = (k << 13) & 0xfff00000) */
    const uint64_t hi = (uint64_t)(((uint32_t)k << 13) & 0xfff00000u) << 32;
    return kd_u2d(kd_d2u(y) + hi);
}
KD_FN double kd_exp_neg(double a)
{
    int k; double r;
    kd_exp_neg_reduce(a, &k, &r);
    const int idx = k & 127;
    return kd_exp_neg_combine(k, kd_exp_neg_poly(r), KD_EXPTAB(2 * idx), KD_EXPTAB(2 * idx + 1));
}

/* log(x) for x in [1, 2] — log(1 + t), t = exp(-|Xp|) in (0, 1], of the logistic rows.  Table method like kd_log_u01 on the one binade
 * the argument can lie in: bin i = the 7 leading mantissa bits (x = 2 joins the last bin), r = x invc_i - 1 exactly rounded, |r| <= 2^-8
 * (bin 0 has c = 1: r = x - 1 exactly, < 2^-7, so log12(1) = 0 and the result keeps its relative accuracy as x -> 1),
 * log x = (logc_i + r) + r^2 (-1/2 + r/3 - r^2/4 + r^3/5 - r^4/6 + r^5/7): no exponent handling, a shorter polynomial (the dropped term
 * r^8/8 < 2^-59), two roundings at the scale of the result: absolute error below 1.5 * 2^-53 on the whole interval
 * (tests/test_oracle_kats.py).  Split at its gather like the others. */
KD_FN uint32_t kd_log12_bin(double x)
{
    const uint32_t i = ((uint32_t)(kd_d2u(x) >> 32) - 0x3ff00000u) >> 13;
    return i < 127u ? i : 127u;
}
KD_FN double kd_log12_finish(double x, double invc, double logc)
{
    const double A2 = -0.5, A3 = 0x1.5555555555555p-2, A4 = -0.25, A5 = 0x1.999999999999ap-3, A6 = -0x1.5555555555555p-3,
                 A7 = 0x1.2492492492492p-3;
    const double r = kd_fma(x, invc, -1.0);
    const double p = kd_fma(r, kd_fma(r, kd_fma(r, kd_fma(r, kd_fma(r, A7, A6), A5), A4), A3), A2);
    return kd_fma(r * r, p, logc + r);
}
KD_FN double kd_log12(double x)
{
    const uint32_t i = kd_log12_bin(x);
    return kd_log12_finish(x, KD_L12TAB(2 * i), KD_L12TAB(2 * i + 1));
}

/* log of a positive number that may be +inf or NaN (1 + exp(.) of the logistic target): the table log plus the two
 * pass-through cases */
KD_FN double kd_log_pos(double x)
{
    double r = kd_log_u01(x < __builtin_inf() ? x : 1.0);
    if (!(x < __builtin_inf())) r = x + x;          /* +inf, NaN */
    return r;
}

/* log(1 + exp(x)) and 1 / (1 + exp(-x)) from ONE exponential, t = exp(-|x|) in [0, 1]:
 *   log(1 + exp(x)) = max(x, 0) + log(1 + t),    1 / (1 + exp(-x)) = x >= 0 ? 1 / (1 + t) : t / (1 + t).
 * The logistic-regression targets need both per data row (doc/examples/swiss/MALA/analytical.jl:13,17 write exp(Xp) and exp(-Xp)
 * separately): one exponential instead of two, and no overflow for large |x| (the literal form gives log(inf) beyond x = 709).
 * Beyond |x| = 708 the pair is (max(x, 0), x >= 0 ? 1 : exp(-708) = 3.3e-308). */
/* n / d for d in [1, 2] and n zero or a normal number in [2^-1021, 1]: the quotient is zero or a normal number and no intermediate
 * can overflow or lose bits to underflow, so the range scaling of the general division (v_div_scale x 2, v_div_fmas, v_div_fixup) has
 * nothing to do — what remains of the compiler's expansion is the reciprocal estimate, two Newton steps, the quotient and one
 * correction fma, correctly rounded like the IEEE division the host takes (8 instead of 11 vector instructions per data row). */
KD_FN double kd_div_unit_range(double n, double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = kd_fma(r, kd_fma(-d, r, 1.0), r);
    r = kd_fma(r, kd_fma(-d, r, 1.0), r);
    const double q = n * r;
    return kd_fma(kd_fma(-d, q, n), r, q);
#else
    return n / d;
#endif
}
KD_FN void kd_softplus_logistic_rows(double x, double* softplus, double* logistic)
{
    const double ax = __builtin_fabs(x);             /* (a source modifier on the device) */
    const double t = kd_exp_neg(ax);                 /* (0, 1], never NaN; exp(-708) beyond |x| = 708 */
    const double onept = 1.0 + t;                    /* [1, 2] */
    const double l1p = kd_log12(onept);
    *softplus = (x > 0.0 ? x : 0.0) + l1p;
    *logistic = kd_div_unit_range(x >= 0.0 ? 1.0 : t, onept);
}
/* The pair as a function of any double: NaN is passed through.  (The data rows of the logistic targets call the form above, which
 * returns finite values for a NaN argument.  A NaN Xp can only come from a non-finite parameter vector.  Every evaluation that forms the
 * log-target also forms the row's term Xp * y, which is then NaN whatever y is, and so is the log-target — what initialize! tests
 * (MALA.jl:83-84) and what every Metropolis test of a proposal reads: the proposal is rejected exactly as the literal arithmetic
 * rejects it.  The one evaluation WITHOUT a log-target is the gradient inside an HMC trajectory: there the rows' part of the gradient
 * comes out finite where the literal X'(y - 1/(1+exp(-Xp))) is NaN in every component, but the component that made Xp non-finite is
 * NaN in the gradient anyway through -p/lambda, stays NaN through the remaining leapfrogs, and the trajectory's closing log-target is
 * NaN: the same rejection.  So the per-row pass-through — a compare and four selects on each of ndata rows — buys nothing observable
 * and is left out) */
KD_FN void kd_softplus_logistic(double x, double* softplus, double* logistic)
{
    kd_softplus_logistic_rows(x, softplus, logistic);
    if (x != x) { *softplus = x; *logistic = x; }
}

/* ---------------------------------------------------------------- erf */
/* erf for the tuner's erf_rate_score (src/tuners/AcceptanceRateMCTuner.jl:17: erf(k x) + 1; Julia's erf is openlibm's = FreeBSD msun
 * s_erf.c).  The evaluation below is that algorithm, operation for operation — the interval split, the rational approximations on
 * |x| < 0.84375, [0.84375, 1.25), [1.25, 1/0.35), [1/0.35, 6), the two-factor exp(-z^2 - 0.5625) exp((z-x)(z+x) + R/S) with z = x
 * truncated to 32 bits of mantissa — with kd_exp for the exponentials, so the reference's erf_rate_score vectors
 * (test/AcceptanceRateMCTuner.jl:13-14) are reproduced bit for bit (tests/test_oracle_kats.py).  The coefficients are those of s_erf.c,
 * whose notice reads:
 * ====================================================
 * Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 * Developed at SunPro, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this
 * software is freely granted, provided that this notice
 * is preserved.
 * ==================================================== */
/* The coefficients sit in one constant table and the polynomials are Horner loops that are NOT unrolled: erf is evaluated once per
 * tuning period by the chains that use the erf score, but its code is part of every tuned kernel, and 60 double constants kept in
 * registers across the transition loop cost the tuned kernels their zero-scratch budget (tests/test_host_api.py).  The loop performs the
 * operations of the nested form c0 + z (c1 + z (c2 + ...)) in the same order.  For the same reason kd_erf is a real call on the device
 * (noinline): its registers are not part of the calling kernel's allocation. */
#if defined(__clang__)
#define KD_NOUNROLL _Pragma("nounroll")
#else
#define KD_NOUNROLL _Pragma("GCC unroll 1")
#endif
KD_FN double kd_horner(const double* c, int n, double z)        /* c[0] + z (c[1] + z (... + z c[n-1])) */
{
    double r = c[n - 1];
    KD_NOUNROLL
    for (int i = n - 2; i >= 0; --i) r = c[i] + z * r;
    return r;
}
#if defined(__HIPCC__)
#define KD_FN_COLD static __host__ __device__ __attribute__((noinline))
#else
#define KD_FN_COLD static __attribute__((noinline))
#endif
KD_FN_COLD double kd_erf(double x)
{
    static const double T[] = {
        /* pp0..pp4  [0]  */ 1.28379167095512558561e-01, -3.25042107247001499370e-01, -2.84817495755985104766e-02,
                             -5.77027029648944159157e-03, -2.37630166566501626084e-05,
        /* 1,qq1..5  [5]  */ 1.0, 3.97917223959155352819e-01, 6.50222499887672944485e-02, 5.08130628187576562776e-03,
                             1.32494738004321644526e-04, -3.96022827877536812320e-06,
        /* pa0..pa6  [11] */ -2.36211856075265944077e-03, 4.14856118683748331666e-01, -3.72207876035701323847e-01,
                             3.18346619901161753674e-01, -1.10894694282396677476e-01, 3.54783043256182359371e-02,
                             -2.16637559486879084300e-03,
        /* 1,qa1..6  [18] */ 1.0, 1.06420880400844228286e-01, 5.40397917702171048937e-01, 7.18286544141962662868e-02,
                             1.26171219808761642112e-01, 1.36370839120290507362e-02, 1.19844998467991074170e-02,
        /* ra0..ra7  [25] */ -9.86494403484714822705e-03, -6.93858572707181764372e-01, -1.05586262253232909814e+01,
                             -6.23753324503260060396e+01, -1.62396669462573470355e+02, -1.84605092906711035994e+02,
                             -8.12874355063065934246e+01, -9.81432934416914548592e+00,
        /* 1,sa1..8  [33] */ 1.0, 1.96512716674392571292e+01, 1.37657754143519042600e+02, 4.34565877475229228821e+02,
                             6.45387271733267880336e+02, 4.29008140027567833386e+02, 1.08635005541779435134e+02,
                             6.57024977031928170135e+00, -6.04244152148580987438e-02,
        /* rb0..rb6  [42] */ -9.86494292470009928597e-03, -7.99283237680523006574e-01, -1.77579549177547519889e+01,
                             -1.60636384855821916062e+02, -6.37566443368389627722e+02, -1.02509513161107724954e+03,
                             -4.83519191608651397019e+02,
        /* 1,sb1..7  [49] */ 1.0, 3.03380607434824582924e+01, 3.25792512996573918826e+02, 1.53672958608443695994e+03,
                             3.19985821950859553908e+03, 2.55305040643316442583e+03, 4.74528541206955367215e+02,
                             -2.24409524465858183362e+01 };
    const double erx = 8.45062911510467529297e-01, efx = 1.28379167095512586316e-01, efx8 = 1.02703333676410069053e+00;
    const double tiny = 1e-300;
    if (x != x) return x;
    const double ax = x < 0.0 ? -x : x;
    const int neg = x < 0.0;
    if (ax < 0.84375) {
        if (ax < 0x1p-28) {
            if (ax < 0x1p-1015) return 0.125 * (8.0 * x + efx8 * x);          /* avoid underflow */
            return x + efx * x;
        }
        const double z = x * x;
        const double y = kd_horner(T + 0, 5, z) / kd_horner(T + 5, 6, z);
        return x + x * y;
    }
    if (ax < 1.25) {
        const double sd = ax - 1.0;
        const double PQ = kd_horner(T + 11, 7, sd) / kd_horner(T + 18, 7, sd);
        return neg ? -erx - PQ : erx + PQ;
    }
    if (ax >= 6.0) return neg ? tiny - 1.0 : 1.0 - tiny;                    /* (also +-inf) */
    const double sd = 1.0 / (ax * ax);
    const int lo = ax < 1.0 / 0.35;
    const double RS = kd_horner(lo ? T + 25 : T + 42, lo ? 8 : 7, sd) / kd_horner(lo ? T + 33 : T + 49, lo ? 9 : 8, sd);
    const double z = kd_u2d(kd_d2u(ax) & 0xffffffff00000000ull);
    const double r = kd_exp(-z * z - 0.5625) * kd_exp((z - ax) * (z + ax) + RS);
    return neg ? r / ax - 1.0 : 1.0 - r / ax;
}

/* ---------------------------------------------------------------- sin/cos(2*pi*u) */
/* The angle is taken from the bits of uu = 1 + m 2^-52 (u = uu - 1 + 2^-53 is the kd_u52 uniform): j = leading 8 bits
 * of m picks the table angle 2 pi (j + 1/2)/256, t = uu - (1 + (j + 1/2)/256) is exact with |t| <= 2^-9, and
 * y = 2 pi (t + 2^-53), |y| <= 0.0123, is rotated onto the table entry:
 *   cos(a + y) = C cos y - S sin y,  sin(a + y) = S cos y + C sin y,  sin y / cos y - 1 by short Taylor polynomials
 * (dropped terms < 2e-20).  Absolute error < 2^-52; no quadrant logic, no division. */
KD_FN void kd_sincos2pi_bits(uint64_t uu_bits, double* sn, double* cs)
{
    const double S1 = -0x1.5555555555555p-3, S2 = 0x1.1111111111111p-7, S3 = -0x1.a01a01a01a01ap-13;
    const double C2 = 0x1.5555555555555p-5, C3 = -0x1.6c16c16c16c17p-10;
    const uint32_t uh = (uint32_t)(uu_bits >> 32);
    const uint32_t j = (uh >> 12) & 255u;
    const double uu = kd_u2d(uu_bits);
    const double cc = kd_u2d((uint64_t)((uh & 0xfffff000u) | 0x00000800u) << 32);
    const double t = uu - cc;
    const double y = kd_fma(t, KD_TWOPI_HI, kd_fma(t, KD_TWOPI_LO, KD_TWOPI_2M53));
    const double z = y * y;
    const double sy = kd_fma(y * z, kd_fma(z, kd_fma(z, S3, S2), S1), y);       /* sin y */
    const double dc = z * kd_fma(z, kd_fma(z, C3, C2), -0.5);                   /* cos y - 1 */
    const double C = KD_SCTAB(2 * j), S = KD_SCTAB(2 * j + 1);
    *cs = kd_fma(-S, sy, kd_fma(C, dc, C));
    *sn = kd_fma(C, sy, kd_fma(S, dc, S));
}
/* double-argument form for u in [0,1): u + (1 - 2^-53) is exact for every kd_u52 value (rounded otherwise) */
KD_FN void kd_sincos2pi(double u, double* sn, double* cs)
{
    kd_sincos2pi_bits(kd_d2u(u + 0x1.fffffffffffffp-1), sn, cs);
}

/* ---------------------------------------------------------------- sqrt of the Box-Muller radicand */
/* y = -2 log(u) lies in [2.2e-16, 73.5]: positive, normal, far from overflow.  The compiler's IEEE f64 sqrt on gfx950
 * is v_rsq_f64 + a Goldschmidt/Newton sequence wrapped in input scaling (ldexp, compare, selects) and a class check for
 * 0/inf/NaN; for this range the wrapper is dead weight (6 of 16 instructions).  Same core sequence, same correctly
 * rounded result as the host's sqrt (tests/test_gpu_parity.py::test_device_math_bit_exact, op 8). */
KD_FN double kd_sqrt_radicand(double y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rsq(y);
    double g = y * r, h = 0.5 * r;
    const double e = kd_fma(-h, g, 0.5);
    g = kd_fma(g, e, g);
    h = kd_fma(h, e, h);
    double d = kd_fma(-g, g, y);
    g = kd_fma(d, h, g);
    d = kd_fma(-g, g, y);
    return kd_fma(d, h, g);
#else
    return __builtin_sqrt(y);
#endif
}

/* ---------------------------------------------------------------- Box-Muller */
/* The normals of the samplers (proposals, momenta, initial states): 64 bits -> two N(0,1),
 *   rad = sqrt(-2 log u1); z0 = rad cos(2 pi u2); z1 = rad sin(2 pi u2).
 * A Philox block carries two such pairs, so the 41 vector instructions of a block are paid once per FOUR normals (rounds 1-3 spent a
 * whole block on a pair: 52-bit u1 and u2).  The lattice underneath:
 *   radius  u1 = kd_u44(wa, wb)            44 bits: u1 >= 2^-45, |z| <= 7.9 (the mass beyond is 3e-15 per draw)
 *   angle   u2 = ((wb >> 12) + 1/2) 2^-20 + 2^-53   20 bits: 2^20 equally spaced directions, the CENTRES of the cells — no direction lies on a
 *                                          coordinate axis (ABI 5 used the cells' left edges: k = 0, 2^18, 2^19, 3 2^18 put mass 2^-19 of
 *                                          every normal within 1e-16 of zero; tests/test_stream_joint.py counts |z| < 1e-9).
 *                                          The marginal of rad cos / rad sin over an equally spaced set of directions is the rectangle
 *                                          rule on a periodic analytic integrand — exact to rounding for smooth test functions; jointly
 *                                          (z0, z1) lie on 2^20 rays with a 44-bit radius along each, so indicator-type statistics see the
 *                                          lattice at the 2^-20 level (the spacing of neighbouring rays at radius r is 6e-6 r)
 * (u1, log u1) are handed back: a layout's padding pair at index ceil(D/2) is the accept draw (kd_accept_uniform) for free. */
KD_FN uint64_t kd_angle_bits20(uint32_t wb)              /* bits of 1 + (k + 1/2) 2^-20: the half cell is bit 31 of the low word, a constant */
{
#if defined(__HIP_DEVICE_COMPILE__)
    return ((uint64_t)__builtin_amdgcn_alignbit(0x3ffu, wb, 12) << 32) | 0x80000000ull;
#else
    return ((uint64_t)(0x3ff00000u | (wb >> 12)) << 32) | 0x80000000ull;
#endif
}
KD_FN void kd_normal_pair_w(uint32_t wa, uint32_t wb, double* z0, double* z1, double* u1_out, double* logu1_out)
{
    const double u1 = kd_u44(wa, wb);
    const double lg = kd_log_u01(u1);
    const double rad = kd_sqrt_radicand(-2.0 * lg);
    double sn, cs;
    kd_sincos2pi_bits(kd_angle_bits20(wb), &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
    *u1_out = u1;
    *logu1_out = lg;
}

/* Which 64 bits: element pair p (elements 2p, 2p + 1 of a D-vector, p < ceil(D/2)) takes half (p >> 3) & 1 — words (x, y) or (z, w) —
 * of block slot (p & 7) + 8 (p >> 4): pairs p and p + 8 share a block.  Both sit in the same lane in every pair-transposed layout
 * (lane q of Q <= 8 holds pairs q + Q j), so the sharing costs no cross-lane traffic.  Slots used: < ceil(D/2).
 * Pair indices at or beyond ceil(D/2) exist as layout padding only; they take words (x, y) of block slot p (their normals are
 * discarded), and index ceil(D/2) itself is the accept draw of the transition: kd_accept_uniform of block slot ceil(D/2). */
KD_FN uint32_t kd_pair_block(uint32_t p) { return (p & 7u) | ((p >> 4) << 3); }
KD_FN uint32_t kd_pair_half(uint32_t p) { return (p >> 3) & 1u; }
KD_FN void kd_normal_pair_at(uint64_t seed, uint64_t chain, uint64_t transition, uint32_t p, uint32_t nreal,
                             double* z0, double* z1, double* u1_out, double* logu1_out)
{
    const int real = p < nreal;
    const int half = real && kd_pair_half(p) != 0u;
    const kd_u32x4 b = kd_stream_block(seed, chain, transition, real ? kd_pair_block(p) : p);
    kd_normal_pair_w(half ? b.z : b.x, half ? b.w : b.y, z0, z1, u1_out, logu1_out);
}
KD_FN double kd_accept_uniform(kd_u32x4 b) { return kd_u44(b.x, b.y); }

/* uniforms of the slice sampler: words (x,y) of a block (or (z,w) for the second), 52 bits */
KD_FN double kd_uniform_xy(kd_u32x4 b) { return kd_u52(b.x, b.y); }
KD_FN double kd_uniform_zw(kd_u32x4 b) { return kd_u52(b.z, b.w); }
/* the uniform of shrink attempt a >= 1 of a coordinate whose draws start at block slot `base`: attempts 2k - 1 and 2k share block base | k
 * (words (x, y), then (z, w)) — a kernel that loops over the attempts forms a block at every odd attempt only */
KD_FN uint32_t kd_slice_attempt_slot(uint32_t base, uint32_t a) { return base | ((a + 1u) >> 1); }
KD_FN double kd_slice_attempt_uniform(uint64_t seed, uint64_t chain, uint64_t transition, uint32_t base, uint32_t a)
{
    const kd_u32x4 b = kd_stream_block(seed, chain, transition, kd_slice_attempt_slot(base, a));
    return (a & 1u) ? kd_uniform_xy(b) : kd_uniform_zw(b);
}

#endif /* KLARA_DETMATH_H */
