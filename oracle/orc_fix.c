/* orc_fix.c -- oracle fixes and force_clear.  TEST INFRASTRUCTURE (see orc.h).
 *
 * Restates
 *   AtomVecUCG::force_clear                UCG/atom_vec_ucg.cpp:131-135 (+ upstream Verlet::force_clear)
 *   FixNVE_UCGLD::initial/final_integrate  UCG/fix_nve_ucgld.cpp:36-153 (per-type mass branch)
 *   Fix_UCGLD_Langevin::init               UCG/fix_ucgld_langevin.cpp:149-183
 *                     ::compute_target     :318-353 (CONSTANT tstyle)
 *                     ::post_force_templated<0>  :226-297
 *                     ::end_of_step        :303-312
 *   FixUCGState::post_force                UCG/fix_ucgstate.cpp:88-132
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_force_clear(orc_atoms *a, int include_ghosts)
{
  const size_t n = (size_t) a->nlocal + (include_ghosts ? (size_t) a->nghost : 0);
  memset(a->f, 0, 3 * n * sizeof(double));
  memset(a->ucgforce, 0, n * sizeof(double));
  memset(a->scores, 0, 2 * n * sizeof(double));
}

void orc_fix_nve_initial(orc_atoms *a, double dt, double ftm2v, int groupbit)
{
  const double dtv = dt;
  const double dtf = 0.5 * dt * ftm2v;
  double dtfm, dtflm;
  double *x = a->x, *v = a->v, *f = a->f;
  for (int i = 0; i < a->nlocal; i++) {
    if (a->mask[i] & groupbit) {
      dtfm = dtf / a->mass[a->type[i]];
      v[3 * i + 0] += dtfm * f[3 * i + 0];
      v[3 * i + 1] += dtfm * f[3 * i + 1];
      v[3 * i + 2] += dtfm * f[3 * i + 2];
      x[3 * i + 0] += dtv * v[3 * i + 0];
      x[3 * i + 1] += dtv * v[3 * i + 1];
      x[3 * i + 2] += dtv * v[3 * i + 2];

      dtflm = dtf / a->ucgml[i];
      a->ucgvl[i] += dtflm * a->ucgforce[i];
      a->ucgl[i] += dtv * a->ucgvl[i];
    }
  }
}

void orc_fix_nve_final(orc_atoms *a, double dt, double ftm2v, int groupbit)
{
  const double dtf = 0.5 * dt * ftm2v;
  double dtfm, dtflm;
  double *v = a->v, *f = a->f;
  for (int i = 0; i < a->nlocal; i++) {
    if (a->mask[i] & groupbit) {
      dtfm = dtf / a->mass[a->type[i]];
      v[3 * i + 0] += dtfm * f[3 * i + 0];
      v[3 * i + 1] += dtfm * f[3 * i + 1];
      v[3 * i + 2] += dtfm * f[3 * i + 2];

      dtflm = dtf / a->ucgml[i];
      a->ucgvl[i] += dtflm * a->ucgforce[i];
    }
  }
}

/* fix nve/ucgld/wall/hard (UCG/fix_nve_ucgld_wall_hard.cpp): the nve/ucgld update, plus
   ucgstate from lambda in initial_integrate (:97-103, :124-130), reflection of lambda at 0 and 1
   after the second half-kick (:171-177, :192-198), and an optional bias force (:223-241). */
void orc_fix_nve_wall_initial(orc_atoms *a, double dt, double ftm2v, int groupbit)
{
  const double dtv = dt;
  const double dtf = 0.5 * dt * ftm2v;
  double dtfm, dtflm;
  double *x = a->x, *v = a->v, *f = a->f;
  for (int i = 0; i < a->nlocal; i++) {
    if (a->mask[i] & groupbit) {
      dtfm = dtf / a->mass[a->type[i]];
      v[3 * i + 0] += dtfm * f[3 * i + 0];
      v[3 * i + 1] += dtfm * f[3 * i + 1];
      v[3 * i + 2] += dtfm * f[3 * i + 2];
      x[3 * i + 0] += dtv * v[3 * i + 0];
      x[3 * i + 1] += dtv * v[3 * i + 1];
      x[3 * i + 2] += dtv * v[3 * i + 2];

      dtflm = dtf / a->ucgml[i];
      a->ucgvl[i] += dtflm * a->ucgforce[i];
      a->ucgl[i] += dtv * a->ucgvl[i];

      if (a->ucgl[i] < 0.5) a->ucgstate[i] = 0;
      else a->ucgstate[i] = 1;
    }
  }
}

void orc_fix_nve_wall_final(orc_atoms *a, double dt, double ftm2v, int groupbit)
{
  const double dtf = 0.5 * dt * ftm2v;
  double dtfm, dtflm;
  double *v = a->v, *f = a->f;
  for (int i = 0; i < a->nlocal; i++) {
    if (a->mask[i] & groupbit) {
      dtfm = dtf / a->mass[a->type[i]];
      v[3 * i + 0] += dtfm * f[3 * i + 0];
      v[3 * i + 1] += dtfm * f[3 * i + 1];
      v[3 * i + 2] += dtfm * f[3 * i + 2];

      dtflm = dtf / a->ucgml[i];
      a->ucgvl[i] += dtflm * a->ucgforce[i];

      if (a->ucgl[i] < 0.0) {
        a->ucgl[i] = -a->ucgl[i];
        a->ucgvl[i] = -a->ucgvl[i];
      } else if (a->ucgl[i] > 1.0) {
        a->ucgl[i] = 2.0 - a->ucgl[i];
        a->ucgvl[i] = -a->ucgvl[i];
      }
    }
  }
}

/* bias_force (:216-221): minus the derivative of (798 x^10 - x^2 + 0.1) * 10 H, x = lambda - 1/2,
   with the products taken left to right as written there */
double orc_wall_bias_force(double lmd, double H)
{
  double x = lmd - 0.5;
  return (-7980 * x * x * x * x * x * x * x * x * x + 2 * x) * 10 * H;
}

void orc_fix_nve_wall_post_force(orc_atoms *a, double barrier, int groupbit)
{
  for (int i = 0; i < a->nlocal; i++)
    if (a->mask[i] & groupbit) a->ucgforce[i] += orc_wall_bias_force(a->ucgl[i], barrier);
}

orc_fix_langevin *orc_fix_langevin_create(int ntypes, double t_start, double t_stop,
                                          double t_period, int seed, int me)
{
  orc_fix_langevin *fx = (orc_fix_langevin *) calloc(1, sizeof(orc_fix_langevin));
  fx->ntypes = ntypes;
  fx->t_start = t_start;
  fx->t_target = t_start;
  fx->t_stop = t_stop;
  fx->t_period = t_period;
  fx->seed = seed;
  fx->gfactor1 = (double *) calloc((size_t) ntypes + 1, sizeof(double));
  fx->gfactor2 = (double *) calloc((size_t) ntypes + 1, sizeof(double));
  orc_ranmars_init(&fx->random, seed + me);
  return fx;
}

void orc_fix_langevin_destroy(orc_fix_langevin *fx)
{
  if (!fx) return;
  free(fx->gfactor1);
  free(fx->gfactor2);
  free(fx);
}

void orc_fix_langevin_init(orc_fix_langevin *fx, const orc_atoms *a, double dt, double boltz,
                           double ftm2v, double mvv2e)
{
  /* :164-171 -- atom->ucgml is a PER-ATOM array indexed here by the TYPE index
     (SURVEY.md App. B #5); reproduced as shipped, ratio[i] = 1 */
  for (int i = 1; i <= fx->ntypes; i++) {
    fx->gfactor1[i] = -a->ucgml[i] / fx->t_period / ftm2v;
    fx->gfactor2[i] = sqrt(a->ucgml[i]) / ftm2v;
    fx->gfactor2[i] *= sqrt(24.0 * boltz / fx->t_period / dt / mvv2e);
    fx->gfactor1[i] *= 1.0 / 1.0;
    fx->gfactor2[i] *= 1.0 / sqrt(1.0);
  }
}

void orc_fix_langevin_post_force(orc_fix_langevin *fx, orc_atoms *a, int groupbit,
                                 long long ntimestep, long long beginstep, long long endstep)
{
  /* compute_target :318-330 */
  double delta = (double) (ntimestep - beginstep);
  if (delta != 0.0) delta /= (double) (endstep - beginstep);
  fx->t_target = fx->t_start + delta * (fx->t_stop - fx->t_start);
  fx->tsqrt = sqrt(fx->t_target);

  double gamma1, gamma2, fdrag, fran;
  for (int i = 0; i < a->nlocal; i++) {
    if (a->mask[i] & groupbit) {
      gamma1 = fx->gfactor1[a->type[i]];
      gamma2 = fx->gfactor2[a->type[i]] * fx->tsqrt;
      fran = gamma2 * (orc_ranmars_uniform(&fx->random) - 0.5);
      fdrag = gamma1 * a->ucgvl[i];
      a->ucgforce[i] += fdrag + fran;
    }
  }
}

void orc_fix_langevin_end_of_step(orc_fix_langevin *fx, const orc_atoms *a, int groupbit,
                                  double boltz, double mvv2e)
{
  double lmd_ek = 0.0;
  for (int i = 0; i < a->nlocal; i++)
    if (a->mask[i] & groupbit) lmd_ek += 0.5 * a->ucgml[i] * a->ucgvl[i] * a->ucgvl[i] * mvv2e;
  fx->lambda_temp = lmd_ek / (0.5 * boltz * a->nlocal);
}

void orc_fix_ucgstate_init(orc_fix_ucgstate *fx, int ld_flag, int mc_flag, int mc_seed,
                           double mc_rate, int me)
{
  fx->ld_flag = ld_flag;
  fx->mc_flag = mc_flag;
  fx->mc_seed = mc_seed;
  fx->mc_rate = mc_rate;
  if (mc_flag) orc_ranmars_init(&fx->random, mc_seed + me);
}

void orc_fix_ucgstate_post_force(orc_fix_ucgstate *fx, orc_atoms *a)
{
  double softmax_denom, mc_factor, mc_rand;
  double ex[2];
  for (int i = 0; i < a->nlocal; i++) {
    if (a->num_ucgstates[i] == 1) {
      if (!fx->ld_flag) a->ucgstate[i] = 0;
      a->ucgp[i] = 1.0;
    } else {
      softmax_denom = 0.0;
      for (int si = 0; si < a->num_ucgstates[i]; si++) {
        double s = a->scores[2 * i + si];
        ex[si] = orc_exp((700.0 < s) ? 700.0 : s); /* std::min(score, 700.0) */
        softmax_denom += ex[si];
      }
      {
        double r = ex[1] / softmax_denom;
        double lo = (1e-6 < r) ? r : 1e-6;               /* std::max(1e-6, r) */
        a->ucgp[i] = (lo < 1.0 - 1e-6) ? lo : 1.0 - 1e-6; /* std::min(1-1e-6, .) */
      }
      if (!fx->ld_flag) {
        if (fx->mc_flag) {
          /* :113-123, rule reproduced exactly (App. B #18) */
          if (a->ucgstate[i] == 0) mc_factor = a->ucgp[i] / (1.0 - a->ucgp[i]);
          else mc_factor = (1.0 - a->ucgp[i]) / a->ucgp[i];
          mc_factor = ((1.0 < mc_factor) ? 1.0 : mc_factor) * fx->mc_rate; /* std::min(mc_factor, 1.0) */
          mc_rand = orc_ranmars_uniform(&fx->random);
          if (mc_rand < mc_factor) a->ucgstate[i] = 0;
          else a->ucgstate[i] = 1;
        } else {
          a->ucgstate[i] = (int) round(a->ucgp[i]);
        }
      }
    }
    if (!fx->ld_flag) a->ucgl[i] = a->ucgp[i];
  }
}
