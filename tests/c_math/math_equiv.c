/* math_equiv.c -- TEST HARNESS: the one place that sees both definitions of "ucg-math-v1".
 *
 * Compares, bit for bit, on pseudo-random arguments spread over every range and every k boundary of the
 * argument reduction:
 *   (a) the product's functions (lammps-ucg-dev_amd/csrc/ucg_math.h: ucg_exp, ucg_expm1, ucg_log, ucg_tanh_branchy)
 *       with the oracle's own (oracle/orc_math.c, reached through orc_exp ... with orc_set_math(0));
 *   (b) the product's branch-light forms the HIP kernels run (ucg_exp_nb, ucg_expm1_nb, ucg_exp_expm1,
 *       ucg_log_nb, ucg_tanh) with the oracle's.
 * Built by tests/test_oracle.py with gcc -O2 -ffp-contract=off and linked with liborc.so.
 * usage: math_equiv N SEED  -> prints "<mismatches a> <mismatches b>"
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../lammps-ucg-dev_amd/csrc/ucg_math.h"

void orc_set_math(int use_libm);
double orc_exp(double x);
double orc_expm1(double x);
double orc_log(double x);
double orc_tanh(double x);

static unsigned long long st;
static unsigned long long rnd(void)
{
  st ^= st << 13;
  st ^= st >> 7;
  st ^= st << 17;
  return st;
}
static int same(double a, double b) { return memcmp(&a, &b, sizeof a) == 0 || (a != a && b != b); }

int main(int argc, char **argv)
{
  const long long n = argc > 1 ? atoll(argv[1]) : 1000000;
  st = argc > 2 ? strtoull(argv[2], 0, 10) : 88172645463325252ull;
  if (!st) st = 88172645463325252ull;
  long long bad_a = 0, bad_b = 0;
  orc_set_math(0);
  /* the thresholds of the definition and their neighbours */
  const double edge[] = {0.34657359027997264, 1.0, 1.0397207708399179, 38.816242111356935, 709.782712893384, 745.1332191019411,
                         5.551115123125783e-17, 2.7755575615628914e-17, 22.0, 0.25, 0.5, 708.0, 44.0, 2.0, 1e-300, 2.2250738585072014e-308};
  for (long long i = 0; i < n; i++) {
    const unsigned long long r = rnd();
    const double u = (double) (r >> 11) / 9007199254740992.0;
    double x;
    switch (i % 10) {
      case 0: x = (u - 0.5) * 1500.0; break;
      case 1: x = (u - 0.5) * 90.0; break;
      case 2: x = (u - 0.5) * 4.0; break;
      case 3: memcpy(&x, &r, sizeof x); break;                /* any bit pattern */
      case 4: x = (u - 0.5) * 1e-15; break;
      case 5: x = ((double) ((long long) (r % 2200) - 1100) + 0.5) * 0.6931471805599453 * (1.0 + (u - 0.5) * 4e-16); break; /* k boundaries */
      case 6: x = 700.0 + u * 12.0; break;
      case 7: {                                               /* a few ulps around a threshold, both signs */
        double e = edge[(r >> 3) % (sizeof edge / sizeof edge[0])];
        long long b;
        memcpy(&b, &e, sizeof b);
        b += (long long) ((r >> 20) % 9) - 4;
        memcpy(&x, &b, sizeof x);
        if (r & 4) x = -x;
        break;
      }
      case 8: x = -745.5 + u * 40.0; break;                   /* subnormal results of exp */
      default: x = (u - 0.5) * 50.0; break;
    }
    const double oe = orc_exp(x), om = orc_expm1(x), ot = orc_tanh(x);
    const double xl = 1.0 + x * 1e-7;
    const double ol = orc_log(x), ol2 = orc_log(xl);
    if (!same(oe, ucg_exp(x)) || !same(om, ucg_expm1(x)) || !same(ot, ucg_tanh_branchy(x)) || !same(ol, ucg_log(x)) ||
        !same(ol2, ucg_log(xl)))
      bad_a++;
    double a, b;
    ucg_exp_expm1(x, &a, &b);
    if (!same(a, oe) || !same(b, om) || !same(ucg_exp_nb(x), oe) || !same(ucg_expm1_nb(x), om) || !same(ucg_tanh(x), ot) ||
        !same(ucg_log_nb(x), ol) || !same(ucg_log_nb(xl), ol2))
      bad_b++;
  }
  printf("%lld %lld\n", bad_a, bad_b);
  return 0;
}
