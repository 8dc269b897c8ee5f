/* ucg_math.h -- exactly specified fp64 elementary functions (exp, expm1, log, tanh).
 *
 * Why this exists: the reference calls libm's std::exp / std::expm1 / std::log /
 * std::tanh (UCG/fix_ucgstate.cpp:102, UCG/pair_table_ucg_bethe.cpp:550-551,
 * UCG/pair_table_ucg_bethe_density.cpp:110,120,306,609,652).  glibc's and the
 * GPU's (ocml) versions are both "< 1 ulp" but not bit-identical to each other,
 * and the posteriors feed back into the dynamics (ucgl = ucgp), so a bitwise
 * reproducible trajectory needs ONE definition evaluated with the same IEEE
 * operations on the host and on gfx950.  These are straight-line restatements
 * of the classic fdlibm algorithms (argument reduction + minimax polynomial),
 * written only with + - * / and integer bit moves, so that with FMA contraction
 * off they produce the same bits under gcc and under hipcc.
 *
 * Every translation unit that includes this header must be compiled with
 * -ffp-contract=off.  tests/test_oracle.py checks each function against libm
 * (<= 1 ulp) so a wrong constant cannot hide behind the shared definition.
 */
#ifndef UCG_MATH_H
#define UCG_MATH_H

#include <stdint.h>

#if defined(__HIP__)
#define UCG_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define UCG_HD static inline
#endif

#if defined(__cplusplus) && defined(__HIP__)
#define UCG_BITS_D2U(d) ((uint64_t)__builtin_bit_cast(unsigned long long, (d)))
#define UCG_BITS_U2D(u) (__builtin_bit_cast(double, (unsigned long long)(u)))
#else
static inline uint64_t ucg_d2u_(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
static inline double ucg_u2d_(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }
#define UCG_BITS_D2U(d) ucg_d2u_(d)
#define UCG_BITS_U2D(u) ucg_u2d_(u)
#endif

/* y * 2^k for finite y in [0.25, 4) and integer k, by exponent arithmetic; two
 * steps near the subnormal range so the final rounding is a single multiply. */
UCG_HD double ucg_scalbn_(double y, int k)
{
  if (k >= -1021) {
    if (k > 1023) { /* only reached for results about to overflow */
      y = UCG_BITS_U2D(UCG_BITS_D2U(y) + ((uint64_t)1023 << 52));
      k -= 1023;
      if (k > 1023) k = 1023;
      return y * UCG_BITS_U2D((uint64_t)(k + 1023) << 52);
    }
    return UCG_BITS_U2D(UCG_BITS_D2U(y) + ((uint64_t)(int64_t)k << 52));
  }
  /* twom1000 = 2^-1000 */
  y = UCG_BITS_U2D(UCG_BITS_D2U(y) + ((uint64_t)(int64_t)(k + 1000) << 52));
  return y * 9.33263618503218878990e-302;
}

UCG_HD double ucg_exp(double x)
{
  const double o_threshold = 7.09782712893383973096e+02;
  const double u_threshold = -7.45133219101941108420e+02;
  const double ln2HI = 6.93147180369123816490e-01;
  const double ln2LO = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;

  if (x != x) return x;
  if (x > o_threshold) return UCG_BITS_U2D((uint64_t)0x7ff0000000000000ULL);
  if (x < u_threshold) return 0.0;

  double hi = x, lo = 0.0;
  int k = 0;
  const double ax = x < 0.0 ? -x : x;
  if (ax > 0.34657359027997264) { /* |x| > 0.5 ln2 */
    k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double t = (double)k;
    hi = x - t * ln2HI;
    lo = t * ln2LO;
    x = hi - lo;
  }
  const double t = x * x;
  const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  return ucg_scalbn_(y, k);
}

UCG_HD double ucg_expm1(double x)
{
  const double o_threshold = 7.09782712893383973096e+02;
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02;
  const double Q2 = 1.58730158725481460165e-03;
  const double Q3 = -7.93650757867487942473e-05;
  const double Q4 = 4.00821782732936239552e-06;
  const double Q5 = -2.01099218183624371326e-07;

  if (x != x) return x;
  if (x > o_threshold) return UCG_BITS_U2D((uint64_t)0x7ff0000000000000ULL);
  if (x < -38.816242111356935) return -1.0; /* x < -56 ln2: exp(x)-1 rounds to -1 */

  const double ax = x < 0.0 ? -x : x;
  double hi, lo, c = 0.0;
  int k = 0;
  if (ax > 0.34657359027997264) {
    k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double t = (double)k;
    hi = x - t * ln2_hi;
    lo = t * ln2_lo;
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (ax < 5.551115123125783e-17) { /* |x| < 2^-54 */
    return x;
  }

  const double hfx = 0.5 * x;
  const double hxs = x * hfx;
  const double r1 = 1.0 + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  double t = 3.0 - r1 * hfx;
  double e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1) {
    if (x < -0.25) return -2.0 * (e - (x + 0.5));
    return 1.0 + 2.0 * (x - e);
  }
  double y;
  if (k <= -2 || k > 56) {
    y = 1.0 - (e - x);
    y = ucg_scalbn_(y, k);
    return y - 1.0;
  }
  if (k < 20) {
    /* t = 1 - 2^-k */
    t = UCG_BITS_U2D(((uint64_t)(0x3ff00000u - (0x200000u >> k))) << 32);
    y = t - (e - x);
    y = ucg_scalbn_(y, k);
  } else {
    /* t = 2^-k */
    t = UCG_BITS_U2D(((uint64_t)(0x3ff - k)) << 52);
    y = x - (e + t);
    y += 1.0;
    y = ucg_scalbn_(y, k);
  }
  return y;
}

/* exp(x) without k-dependent control flow for -708 < x <= 709.78 (the rest -- NaN, overflow, the subnormal results --
 * through ucg_exp): |x| <= ln2/2 is the general formula with k = 0 (see ucg_exp_expm1), the scaling by 2^k is exponent
 * arithmetic.  Bit for bit ucg_exp (tests/test_oracle.py). */
UCG_HD double ucg_exp_nb(double x0)
{
  const double ln2HI = 6.93147180369123816490e-01;
  const double ln2LO = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;
  if (!(x0 > -708.0 && x0 <= 7.09782712893383973096e+02)) return ucg_exp(x0);
  const double ax = x0 < 0.0 ? -x0 : x0;
  const int k = (ax > 0.34657359027997264) ? (int)(invln2 * x0 + (x0 < 0.0 ? -0.5 : 0.5)) : 0; /* -1021 ... 1024 */
  const double t = (double)k;
  const double hi = x0 - t * ln2HI;
  const double lo = t * ln2LO;
  const double x = hi - lo;
  const double tt = x * x;
  const double c = x - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k > 1023) return ucg_scalbn_(y, k); /* x within 0.35 of the overflow threshold */
  return UCG_BITS_U2D(UCG_BITS_D2U(y) + ((uint64_t)(int64_t)k << 52));
}

/* a / b for operands the caller knows to lie where the hardware division's operand scaling (v_div_scale_f64) is the
 * identity: b normal with an exponent far from both ends, |a| >= 2^-900 or a == 0, a / b normal.  On gfx950 `a / b` is
 * rcp + two Newton-Raphson steps on the reciprocal + one on the quotient, wrapped in two v_div_scale_f64, v_div_fmas_f64 and
 * v_div_fixup_f64 that rescale extreme operands and patch the special cases; for operands in the range above those four are
 * no-ops, so the bare core returns the same bits (the correctly rounded quotient) with four instructions less.  On the host
 * it IS a / b.  tests/c_math/math_equiv.c and ucg_selftest_div compare the two on the ranges used. */
#if defined(__HIP_DEVICE_COMPILE__)
UCG_HD double ucg_div_core(double a, double b)
{
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, y, q);
}
#else
UCG_HD double ucg_div_core(double a, double b) { return a / b; }
#endif

/* sqrt(x) for x the caller knows to be a normal number >= 2^-767 (finite): on gfx950 `sqrt(x)` is v_rsq_f64 + the coupled
 * Goldschmidt / Newton iteration below, wrapped in an operand scaling (compare, select, two v_ldexp_f64) for x < 2^-767 and a
 * class test with two selects for 0 / inf / NaN; for x in the range above the wrapping is the identity, so the bare core
 * returns the same bits -- the correctly rounded root -- with eleven instructions less.  On the host it IS sqrt(x).
 * ucg_selftest_sqrt_core compares the two on the device. */
#if defined(__HIP_DEVICE_COMPILE__)
UCG_HD double ucg_sqrt_core(double x)
{
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  double d = __builtin_fma(-g, g, x);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
#else
UCG_HD double ucg_sqrt_core(double x) { return sqrt(x); }
#endif

/* exp(x) and expm1(x) of the SAME argument (the Bethe closure needs both, UCG/pair_table_ucg_bethe.cpp:550-551), with
 * one shared argument reduction and almost no control flow -- on a GPU the two functions' many early exits and
 * k-dependent formulas make a wavefront run every path one of its 64 lanes takes.  Every result is bit for bit that of
 * ucg_exp / ucg_expm1 above (tests/test_oracle.py compares them on millions of arguments, all branches included):
 *   - |x| <= ln2/2 is the general path with k = 0: then t = 0, hi = x - 0 = x, lo = 0, c = 0, and the k = 0 formulas
 *     are the general ones ((x c)/(c - 2) = -((x c)/(2 - c)) exactly; scaling by 2^0 adds 0 to the exponent);
 *   - expm1's k = 0 and k = -1 results are formed from the shared x - e and selected; the other k (|x| > 1.04) and the
 *     special arguments (NaN, overflow, underflow, tiny) go through the original functions in one rarely taken branch. */
UCG_HD double ucg_expm1_nb(double x0);

UCG_HD void ucg_exp_expm1(double x0, double *ex, double *em1)
{
  const double ln2HI = 6.93147180369123816490e-01;
  const double ln2LO = 1.90821492927058770002e-10;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;
  const double Q1 = -3.33333333333331316428e-02;
  const double Q2 = 1.58730158725481460165e-03;
  const double Q3 = -7.93650757867487942473e-05;
  const double Q4 = 4.00821782732936239552e-06;
  const double Q5 = -2.01099218183624371326e-07;

  const double ax = x0 < 0.0 ? -x0 : x0;
  /* common range: -1.03972077083991796 < x < ... i.e. k in {-1, 0} and no special case; NaN fails the tests */
  const int common = (ax >= 5.551115123125783e-17) && (x0 > -1.0397207708399179) && (x0 <= 0.34657359027997264);
  /* Outside it (in the benchmark melt: pairs closer than ~0.93 sigma, 0.4 % of the pairs -- but a QUARTER of the wavefronts
     hold at least one): the fixed-sequence forms ucg_exp_nb / ucg_expm1_nb, each bit for bit its original for every argument,
     instead of the originals' k-dependent control flow (on the GPU a wavefront ran every branch some lane needed: ~350
     instructions).  On the device the choice is made per WAVEFRONT, so that a wavefront runs one of the two sequences. */
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_amdgcn_ballot_w64(!common) != 0ull) { /* some active lane of the wavefront is outside the common range */
#else
  if (!common) {
#endif
    *ex = ucg_exp_nb(x0);
    *em1 = ucg_expm1_nb(x0);
    return;
  }
  /* k = 0 for |x| <= ln2/2, else (here x < 0) k = (int)(invln2 x - 0.5) = -1 */
  const int km1 = ax > 0.34657359027997264;
  const double t = km1 ? -1.0 : 0.0;
  const double hi = x0 - t * ln2HI;
  const double lo = t * ln2LO;
  const double x = hi - lo;
  const double c2 = (hi - x) - lo; /* expm1's correction term; 0 when k = 0 */
  /* exp */
  const double tt = x * x;
  const double c = x - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  /* the two quotients below: denominators in [1.3, 2.7] and [5.4, 6.6]; numerators x c ~ x^2 >= 2^-109 and r1 - t3 ~ x / 2
     (or exactly 0): inside ucg_div_core's range for every argument of the common path */
  const double y = 1.0 - ((lo - ucg_div_core(x * c, 2.0 - c)) - hi);
  /* y * 2^k, k in {-1, 0}: exponent arithmetic as ucg_scalbn_ does for k >= -1021 */
  *ex = km1 ? UCG_BITS_U2D(UCG_BITS_D2U(y) - ((uint64_t)1 << 52)) : y;
  /* expm1 */
  const double hfx = 0.5 * x;
  const double hxs = x * hfx;
  const double r1 = 1.0 + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  const double t3 = 3.0 - r1 * hfx;
  double e = hxs * ucg_div_core(r1 - t3, 6.0 - x * t3);
  /* k = 0: x - (x e - hxs);  k = -1: e = x (e - c) - c; e -= hxs; 0.5 (x - e) - 0.5.  With c = 0 the k = -1 chain
     gives e = x e - hxs as well, so both cases share it */
  e = (x * (e - c2) - c2);
  e -= hxs;
  const double d = x - e;
  *em1 = km1 ? 0.5 * d - 0.5 : d;
}

UCG_HD double ucg_log(double x)
{
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double two54 = 1.80143985094819840000e+16;
  const double Lg1 = 6.666666666666735130e-01;
  const double Lg2 = 3.999999999940941908e-01;
  const double Lg3 = 2.857142874366239149e-01;
  const double Lg4 = 2.222219843214978396e-01;
  const double Lg5 = 1.818357216161805012e-01;
  const double Lg6 = 1.531383769920937332e-01;
  const double Lg7 = 1.479819860511658591e-01;

  uint64_t u = UCG_BITS_D2U(x);
  int32_t hx = (int32_t)(u >> 32);
  uint32_t lx = (uint32_t)u;
  int k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return UCG_BITS_U2D((uint64_t)0xfff0000000000000ULL); /* -inf */
    if (hx < 0) return UCG_BITS_U2D((uint64_t)0x7ff8000000000000ULL);                        /* nan  */
    k -= 54;
    x *= two54;
    u = UCG_BITS_D2U(x);
    hx = (int32_t)(u >> 32);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  u = (u & 0xffffffffULL) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
  x = UCG_BITS_U2D(u);
  k += (i >> 20);
  const double f = x - 1.0;
  const double dk = (double)k;
  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2^-20 */
    if (f == 0.0) {
      if (k == 0) return 0.0;
      return dk * ln2_hi + dk * ln2_lo;
    }
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  i = hx - 0x6147a;
  const double w = z * z;
  const int32_t j = 0x6b851 - hx;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  const double R = t2 + t1;
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* log(x) with one fixed instruction sequence for normal positive x not within 2^-20 of a power of two's 1.0 image (the
 * rest -- zero, negative, subnormal, inf, NaN, |f| < 2^-20 -- through ucg_log): k = 0 is the general formula with dk = 0,
 * and fdlibm's two polynomial combinations (i > 0 or not) are both formed and one is selected.  Bit for bit ucg_log. */
UCG_HD double ucg_log_nb(double x0)
{
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01;
  const double Lg2 = 3.999999999940941908e-01;
  const double Lg3 = 2.857142874366239149e-01;
  const double Lg4 = 2.222219843214978396e-01;
  const double Lg5 = 1.818357216161805012e-01;
  const double Lg6 = 1.531383769920937332e-01;
  const double Lg7 = 1.479819860511658591e-01;
  uint64_t u = UCG_BITS_D2U(x0);
  int32_t hx = (int32_t)(u >> 32);
  if (hx < 0x00100000 || hx >= 0x7ff00000) return ucg_log(x0);
  int k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  if ((0x000fffff & (2 + hx)) < 3) return ucg_log(x0); /* |f| < 2^-20 */
  int32_t i = (hx + 0x95f64) & 0x100000;
  u = (u & 0xffffffffULL) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
  const double x = UCG_BITS_U2D(u);
  k += (i >> 20);
  const double f = x - 1.0;
  const double dk = (double)k;
  const double s = f / (2.0 + f);
  const double z = s * s;
  i = hx - 0x6147a;
  const double w = z * z;
  const int32_t j = 0x6b851 - hx;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double ra = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  const double rb = dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
  return (i > 0) ? ra : rb;
}

/* expm1(x) without k-dependent control flow, for 2^-54 <= |x| < 44 (anything else goes through ucg_expm1 in one rarely
 * taken branch): the reduction is done with k = 0 for |x| <= ln2/2 (then hi = x, lo = 0, c = 0), ALL of fdlibm's
 * result formulas (k = 0, -1, 1, |k| large, k < 20, k >= 20) are formed from the shared e and one of them is selected.
 * On a GPU the lanes of a wavefront take different k, and the branchy form runs every formula a lane needs one after
 * the other, masks and scalar control flow included; this one runs a fixed instruction sequence.  Every result is bit
 * for bit that of ucg_expm1 (tests/test_oracle.py).  Used by ucg_tanh, whose arguments spread over k = -3 ... 63. */
UCG_HD double ucg_expm1_nb(double x0)
{
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02;
  const double Q2 = 1.58730158725481460165e-03;
  const double Q3 = -7.93650757867487942473e-05;
  const double Q4 = 4.00821782732936239552e-06;
  const double Q5 = -2.01099218183624371326e-07;

  const double ax = x0 < 0.0 ? -x0 : x0;
  if (!(ax >= 5.551115123125783e-17 && ax < 44.0) || x0 < -38.816242111356935) return ucg_expm1(x0);
  const int k = (ax > 0.34657359027997264) ? (int)(invln2 * x0 + (x0 < 0.0 ? -0.5 : 0.5)) : 0; /* -56 ... 64 */
  const double tk = (double)k;
  const double hi = x0 - tk * ln2_hi;
  const double lo = tk * ln2_lo;
  const double x = hi - lo;
  const double c = (hi - x) - lo;
  const double hfx = 0.5 * x;
  const double hxs = x * hfx;
  const double r1 = 1.0 + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  const double t3 = 3.0 - r1 * hfx;
  double e = hxs * ((r1 - t3) / (6.0 - x * t3));
  e = (x * (e - c) - c);
  e -= hxs;
  const double d = x - e;   /* k = 0: x - (x e - hxs) */
  const double g = e - x;
  const uint64_t kbits = (uint64_t)(int64_t)k << 52; /* scaling by 2^k: exponent arithmetic (|k| <= 64, y in [0.25, 4)) */
  /* k = -1 */
  const double rm1 = 0.5 * d - 0.5;
  /* k = 1 */
  const double r1a = -2.0 * (e - (x + 0.5));
  const double r1b = 1.0 + 2.0 * d;
  const double rp1 = (x < -0.25) ? r1a : r1b;
  /* k <= -2 or k > 56 */
  const double yA = 1.0 - g;
  const double rA = UCG_BITS_U2D(UCG_BITS_D2U(yA) + kbits) - 1.0;
  /* 2 <= k < 20: t = 1 - 2^-k */
  const int kb = k < 0 ? 0 : (k > 31 ? 31 : k);
  const double tB = UCG_BITS_U2D(((uint64_t)(0x3ff00000u - (0x200000u >> kb))) << 32);
  const double yB = tB - g;
  const double rB = UCG_BITS_U2D(UCG_BITS_D2U(yB) + kbits);
  /* 20 <= k <= 56: t = 2^-k */
  const int kc = k < 0 ? 0 : k;
  const double tC = UCG_BITS_U2D(((uint64_t)(0x3ff - kc)) << 52);
  double yC = x - (e + tC);
  yC += 1.0;
  const double rC = UCG_BITS_U2D(UCG_BITS_D2U(yC) + kbits);
  double r = (k < 20) ? rB : rC;
  r = (k <= -2 || k > 56) ? rA : r;
  r = (k == 1) ? rp1 : r;
  r = (k == -1) ? rm1 : r;
  r = (k == 0) ? d : r;
  return r;
}

/* tanh as fdlibm forms it from expm1 (the statements of the original, kept below as ucg_tanh_branchy for the regression
 * test), with the two |x| ranges sharing one division: z = 1 - 2/(t + 2) for |x| >= 1 (t = expm1(2|x|)), z = -t/(t + 2)
 * below (t = expm1(-2|x|)); same bits. */
UCG_HD double ucg_tanh(double x)
{
  const double ax = x < 0.0 ? -x : x;
  if (!(ax >= 2.7755575615628914e-17 && ax < 22.0)) {
    if (x != x) return x;
    if (ax < 22.0) return x * (1.0 + x); /* |x| < 2^-55 */
    return x < 0.0 ? -1.0 : 1.0;
  }
  const int big = ax >= 1.0;
  const double t = ucg_expm1_nb(big ? 2.0 * ax : -2.0 * ax);
  const double q = (big ? 2.0 : -t) / (t + 2.0);
  const double z = big ? 1.0 - q : q;
  return x < 0.0 ? -z : z;
}

UCG_HD double ucg_tanh_branchy(double x)
{
  if (x != x) return x;
  const double ax = x < 0.0 ? -x : x;
  double z;
  if (ax < 22.0) {
    if (ax < 2.7755575615628914e-17) return x * (1.0 + x); /* |x| < 2^-55 */
    if (ax >= 1.0) {
      const double t = ucg_expm1(2.0 * ax);
      z = 1.0 - 2.0 / (t + 2.0);
    } else {
      const double t = ucg_expm1(-2.0 * ax);
      z = -t / (t + 2.0);
    }
  } else {
    z = 1.0;
  }
  return x < 0.0 ? -z : z;
}

#endif /* UCG_MATH_H */
