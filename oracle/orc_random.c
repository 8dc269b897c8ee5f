/* orc_random.c -- oracle RNGs and math selection.  TEST INFRASTRUCTURE (see orc.h).
 *
 * RanMars / RanPark are upstream LAMMPS classes (random_mars.h / random_park.h,
 * absent from /root/reference).  The reference's call sites are
 *   UCG/fix_ucgld_langevin.cpp:85,280   RanMars(seed + me), uniform() per atom
 *   UCG/fix_ucgstate.cpp:62,117         RanMars(mc_seed + me)
 *   UCG/pair_table_ucg_bethe.cpp:187,235,856
 *   UCG/fix_cluster_switch.cpp:56-57,915  RanPark
 * Restated from the published algorithm: Marsaglia, Zaman & Tsang, "Toward a
 * universal random number generator", Stat. Prob. Lett. 9 (1990) 35 (RANMAR),
 * with LAMMPS' seeding convention ij=(seed-1)/30082, kl=(seed-1)-30082*ij and
 * its one warm-up draw in the constructor.  Pinned by the
 * paper's check values in tests/test_oracle_ranmars.py.
 */
#include "orc.h"

#include <math.h>
#include <string.h>

#include "../lammps-ucg-dev_amd/csrc/ucg_math.h"

void orc_ranmars_init(orc_ranmars *r, int seed)
{
  int ij = (seed - 1) / 30082;
  int kl = (seed - 1) - 30082 * ij;
  int i = (ij / 177) % 177 + 2;
  int j = ij % 177 + 2;
  int k = (kl / 169) % 178 + 1;
  int l = kl % 169;
  for (int ii = 1; ii <= 97; ii++) {
    double s = 0.0;
    double t = 0.5;
    for (int jj = 1; jj <= 24; jj++) {
      int m = ((i * j) % 179) * k % 179;
      i = j;
      j = k;
      k = m;
      l = (53 * l + 1) % 169;
      if ((l * m) % 64 >= 32) s += t;
      t *= 0.5;
    }
    r->u[ii] = s;
  }
  r->c = 362436.0 / 16777216.0;
  r->cd = 7654321.0 / 16777216.0;
  r->cm = 16777213.0 / 16777216.0;
  r->i97 = 97;
  r->j97 = 33;
  orc_ranmars_uniform(r);
}

double orc_ranmars_uniform(orc_ranmars *r)
{
  double uni = r->u[r->i97] - r->u[r->j97];
  if (uni < 0.0) uni += 1.0;
  r->u[r->i97] = uni;
  r->i97--;
  if (r->i97 == 0) r->i97 = 97;
  r->j97--;
  if (r->j97 == 0) r->j97 = 97;
  r->c -= r->cd;
  if (r->c < 0.0) r->c += r->cm;
  uni -= r->c;
  if (uni < 0.0) uni += 1.0;
  return uni;
}

void orc_ranmars_fill(orc_ranmars *r, int n, double *out)
{
  for (int i = 0; i < n; i++) out[i] = orc_ranmars_uniform(r);
}

/* Park & Miller minimal standard generator (Schrage form), as LAMMPS RanPark */
void orc_ranpark_init(orc_ranpark *r, int seed) { r->seed = seed; }

double orc_ranpark_uniform(orc_ranpark *r)
{
  int k = r->seed / 127773;
  r->seed = 16807 * (r->seed - k * 127773) - 2836 * k;
  if (r->seed < 0) r->seed += 2147483647;
  return (1.0 / 2147483647.0) * r->seed;
}

static int g_use_libm = 0;
void orc_set_math(int use_libm) { g_use_libm = use_libm; }
double orc_exp(double x) { return g_use_libm ? exp(x) : ucg_exp(x); }
double orc_expm1(double x) { return g_use_libm ? expm1(x) : ucg_expm1(x); }
double orc_log(double x) { return g_use_libm ? log(x) : ucg_log(x); }
double orc_tanh(double x) { return g_use_libm ? tanh(x) : ucg_tanh_branchy(x); } /* the fdlibm-shaped original */

/* the branch-light forms the HIP kernels use (ucg_exp_nb, ucg_expm1_nb, ucg_exp_expm1, ucg_log_nb, ucg_tanh) against the originals
   the oracle uses (ucg_exp, ucg_expm1, ucg_log, ucg_tanh_branchy): number of arguments, out of n pseudo-random ones spread over
   every range and every k boundary of the argument reduction, on which any of them differs in any bit (must be 0) */
static unsigned long long sc_state;
static unsigned long long sc_rnd(void)
{
  sc_state ^= sc_state << 13;
  sc_state ^= sc_state >> 7;
  sc_state ^= sc_state << 17;
  return sc_state;
}
static int sc_same(double a, double b) { return memcmp(&a, &b, sizeof a) == 0 || (a != a && b != b); }
long long orc_math_selfcheck(long long n, unsigned long long seed)
{
  long long bad = 0;
  sc_state = seed ? seed : 88172645463325252ull;
  for (long long i = 0; i < n; i++) {
    const unsigned long long r = sc_rnd();
    const double u = (double) (r >> 11) / 9007199254740992.0;
    double x;
    switch (i % 8) {
      case 0: x = (u - 0.5) * 1500.0; break;
      case 1: x = (u - 0.5) * 90.0; break;
      case 2: x = (u - 0.5) * 4.0; break;
      case 3: memcpy(&x, &r, sizeof x); break;
      case 4: x = (u - 0.5) * 1e-15; break;
      case 5: x = ((double) ((long long) (r % 2200) - 1100) + 0.5) * 0.6931471805599453 * (1.0 + (u - 0.5) * 4e-16); break; /* k boundaries */
      case 6: x = 700.0 + u * 12.0; break;
      default: x = (u - 0.5) * 50.0; break;
    }
    double a, b;
    ucg_exp_expm1(x, &a, &b);
    if (!sc_same(a, ucg_exp(x)) || !sc_same(b, ucg_expm1(x)) || !sc_same(ucg_exp_nb(x), ucg_exp(x)) ||
        !sc_same(ucg_expm1_nb(x), ucg_expm1(x)) || !sc_same(ucg_tanh(x), ucg_tanh_branchy(x)) ||
        !sc_same(ucg_log_nb(x), ucg_log(x)) || !sc_same(ucg_log_nb(1.0 + x * 1e-7), ucg_log(1.0 + x * 1e-7)))
      bad++;
  }
  return bad;
}
