/* orc_random.c -- oracle RNGs.  TEST INFRASTRUCTURE (see orc.h).
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
 * paper's check values in tests/test_oracle.py.
 */
#include "orc.h"

#include <math.h>
#include <string.h>

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
