/* orc_math.c -- the oracle's OWN fp64 exp / expm1 / log / tanh.  TEST INFRASTRUCTURE (see orc.h).
 *
 * The reference calls libm: std::exp / std::expm1 (UCG/pair_table_ucg_bethe.cpp:550-551),
 * std::exp (UCG/fix_ucgstate.cpp:102-107), std::tanh / std::log / std::exp
 * (UCG/pair_table_ucg_bethe_density.cpp:110,120,306,609,652).  libm builds agree with each other only to
 * "< 1 ulp", and the posteriors feed back into the dynamics, so bit-for-bit trajectory tests need ONE
 * written definition of these four functions.  That definition ("ucg-math-v1") is Sun's fdlibm 5.3
 * (e_exp.c, s_expm1.c, e_log.c, s_tanh.c: argument reduction x = k ln2 + r with the two-piece ln2, the
 * minimax polynomials P1..P5 / Q1..Q5 / Lg1..Lg7), evaluated in IEEE double with no fused operations, with
 * these stated details:
 *   - the reduction thresholds 0.5 ln2, 56 ln2 and the overflow / underflow thresholds are compared as
 *     DOUBLES (fdlibm compares high words);
 *   - exp has no tiny-argument shortcut (|x| < 2^-28 runs the k = 0 formula);
 *   - scaling by 2^k is exponent arithmetic, in two steps below 2^-1021 (as fdlibm) and above 2^1023.
 * This file is written from that definition independently of the product: it includes nothing from
 * lammps-ucg-dev_amd/, so the functions on the two sides of every GPU parity test are separate code.
 * tests/test_oracle.py compares the two implementations bit for bit on millions of arguments
 * (tests/c_math/math_equiv.c is the harness that sees both) and each against libm (<= 1 ulp; tanh <= 4).
 * orc_set_math(1) switches the oracle to libm itself -- what the reference runs -- and the GPU tests bound
 * the difference on forces, scores and state trajectories (tests/test_gpu_libm.py).
 */
#include "orc.h"

#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- word access, fdlibm style */
static uint32_t hi_word(double x)
{
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  return (uint32_t) (u >> 32);
}
static uint32_t lo_word(double x)
{
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  return (uint32_t) u;
}
static double with_hi_word(double x, uint32_t hw)
{
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  u = (u & 0xffffffffull) | ((uint64_t) hw << 32);
  memcpy(&x, &u, sizeof x);
  return x;
}
static double from_words(uint32_t hw, uint32_t lw)
{
  const uint64_t u = ((uint64_t) hw << 32) | lw;
  double x;
  memcpy(&x, &u, sizeof x);
  return x;
}

static const double LN2_HI = 6.93147180369123816490e-01; /* 0x3fe62e42, 0xfee00000 */
static const double LN2_LO = 1.90821492927058770002e-10; /* 0x3dea39ef, 0x35793c76 */
static const double INV_LN2 = 1.44269504088896338700e+00;
static const double HALF_LN2 = 0.34657359027997264;      /* 0x3fd62e42, 0xfefa39ef */
static const double EXP_OVERFLOW = 7.09782712893383973096e+02;
static const double EXP_UNDERFLOW = -7.45133219101941108420e+02;

/* y * 2^k, y finite and of order one: add k to the exponent field; results below the normal range take fdlibm's
   detour through 2^(k+1000) * 2^-1000 (one rounding), results at the top of the range the mirror image of it */
static double scale_pow2(double y, int k)
{
  if (k < -1021) {
    y = with_hi_word(y, hi_word(y) + ((uint32_t) (k + 1000) << 20));
    return y * 9.33263618503218878990e-302; /* 2^-1000 */
  }
  if (k > 1023) {
    y = with_hi_word(y, hi_word(y) + ((uint32_t) 1023 << 20));
    k -= 1023;
    if (k > 1023) k = 1023;
    return y * from_words((uint32_t) (k + 1023) << 20, 0);
  }
  return with_hi_word(y, hi_word(y) + ((uint32_t) k << 20));
}

/* e_exp.c */
static double fd_exp(double x)
{
  static const double P[5] = {1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05,
                              -1.65339022054652515390e-06, 4.13813679705723846039e-08};
  if (isnan(x)) return x;
  if (x > EXP_OVERFLOW) return INFINITY;
  if (x < EXP_UNDERFLOW) return 0.0;
  const int neg = x < 0.0;
  double hi = 0.0, lo = 0.0;
  int k = 0;
  if (fabs(x) > HALF_LN2) {
    if (fabs(x) < 1.0) { /* inside 1.5 ln2: k = +-1 without the multiplication (fdlibm's first branch) */
      k = neg ? -1 : 1;
      hi = neg ? x + LN2_HI : x - LN2_HI;
      lo = neg ? -LN2_LO : LN2_LO;
    } else {
      k = (int) (INV_LN2 * x + (neg ? -0.5 : 0.5));
      hi = x - (double) k * LN2_HI;
      lo = (double) k * LN2_LO;
    }
    x = hi - lo;
  }
  const double t = x * x;
  const double c = x - t * (P[0] + t * (P[1] + t * (P[2] + t * (P[3] + t * P[4]))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  return scale_pow2(y, k);
}

/* s_expm1.c */
static double fd_expm1(double x)
{
  static const double Q[5] = {-3.33333333333331316428e-02, 1.58730158725481460165e-03, -7.93650757867487942473e-05,
                              4.00821782732936239552e-06, -2.01099218183624371326e-07};
  if (isnan(x)) return x;
  if (x > EXP_OVERFLOW) return INFINITY;
  if (x < -38.816242111356935) return -1.0; /* below -56 ln2 the result rounds to -1 */
  const int neg = x < 0.0;
  double c = 0.0;
  int k = 0;
  if (fabs(x) > HALF_LN2) {
    double hi, lo;
    if (fabs(x) < 1.0) {
      k = neg ? -1 : 1;
      hi = neg ? x + LN2_HI : x - LN2_HI;
      lo = neg ? -LN2_LO : LN2_LO;
    } else {
      k = (int) (INV_LN2 * x + (neg ? -0.5 : 0.5));
      hi = x - (double) k * LN2_HI;
      lo = (double) k * LN2_LO;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (fabs(x) < 5.551115123125783e-17) { /* 2^-54 */
    return x;
  }
  const double hfx = 0.5 * x;
  const double hxs = x * hfx;
  const double r1 = 1.0 + hxs * (Q[0] + hxs * (Q[1] + hxs * (Q[2] + hxs * (Q[3] + hxs * Q[4]))));
  double t = 3.0 - r1 * hfx;
  double e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = x * (e - c) - c;
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1) return (x < -0.25) ? -2.0 * (e - (x + 0.5)) : 1.0 + 2.0 * (x - e);
  if (k <= -2 || k > 56) return scale_pow2(1.0 - (e - x), k) - 1.0;
  if (k < 20) {
    t = from_words(0x3ff00000u - (0x200000u >> k), 0); /* 1 - 2^-k */
    return scale_pow2(t - (e - x), k);
  }
  t = from_words((uint32_t) (0x3ff - k) << 20, 0); /* 2^-k */
  double y = x - (e + t);
  y += 1.0;
  return scale_pow2(y, k);
}

/* e_log.c */
static double fd_log(double x)
{
  static const double Lg[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
                               2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
                               1.479819860511658591e-01};
  int32_t hx = (int32_t) hi_word(x);
  int k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lo_word(x)) == 0) return -INFINITY;
    if (hx < 0) return from_words(0x7ff80000u, 0);
    k = -54;
    x *= 1.80143985094819840000e+16; /* 2^54: subnormal -> normal */
    hx = (int32_t) hi_word(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  x = with_hi_word(x, (uint32_t) (hx | (i ^ 0x3ff00000))); /* x or x/2, in [sqrt(1/2), sqrt(2)) */
  k += i >> 20;
  const double f = x - 1.0;
  const double dk = (double) k;
  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2^-20 */
    if (f == 0.0) return k == 0 ? 0.0 : dk * LN2_HI + dk * LN2_LO;
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    return k == 0 ? f - R : dk * LN2_HI - ((R - dk * LN2_LO) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg[1] + w * (Lg[3] + w * Lg[5]));
  const double t2 = z * (Lg[0] + w * (Lg[2] + w * (Lg[4] + w * Lg[6])));
  const double R = t2 + t1;
  i = (hx - 0x6147a) | (0x6b851 - hx);
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
  }
  return k == 0 ? f - s * (f - R) : dk * LN2_HI - ((s * (f - R) - dk * LN2_LO) - f);
}

/* s_tanh.c */
static double fd_tanh(double x)
{
  if (isnan(x)) return x;
  const uint32_t ix = hi_word(x) & 0x7fffffffu;
  double z = 1.0;
  if (ix < 0x40360000u) {                          /* |x| < 22 */
    if (ix < 0x3c800000u) return x * (1.0 + x);    /* |x| < 2^-55 */
    if (ix >= 0x3ff00000u) {
      const double t = fd_expm1(2.0 * fabs(x));
      z = 1.0 - 2.0 / (t + 2.0);
    } else {
      const double t = fd_expm1(-2.0 * fabs(x));
      z = -t / (t + 2.0);
    }
  }
  return signbit(x) ? -z : z;
}

static int g_use_libm = 0;
void orc_set_math(int use_libm) { g_use_libm = use_libm; }
int orc_get_math(void) { return g_use_libm; }
double orc_exp(double x) { return g_use_libm ? exp(x) : fd_exp(x); }
double orc_expm1(double x) { return g_use_libm ? expm1(x) : fd_expm1(x); }
double orc_log(double x) { return g_use_libm ? log(x) : fd_log(x); }
double orc_tanh(double x) { return g_use_libm ? tanh(x) : fd_tanh(x); }
