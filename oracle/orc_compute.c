/* orc_compute.c -- oracle compute() for table_ucgld and table_ucg_bethe.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * Two summation orders for the same per-pair arithmetic:
 *
 *  orc_pair_compute_half   "reference order": the reference's own loop -- a
 *      sequential sweep of a HALF list, adding to i and scattering to j
 *      (UCG/pair_table_ucgld.cpp:184-539, UCG/pair_table_ucg_bethe.cpp:165-626),
 *      newton_pair on.  Ghost rows then need a reverse sum into their owners.
 *
 *  orc_pair_compute_gather "canonical order": every owned atom gathers over its
 *      FULL list row, in row order, starting from the same initial value as the
 *      reference's prologue.  Each pair is evaluated in the reference's (i,j)
 *      orientation (bit 29 of the entry says whether the row owner is "i"), so
 *      a pair contributes exactly the numbers the half-list sweep would add to
 *      that atom; only the ORDER of the additions differs.  This is the order
 *      the HIP kernels reproduce bit for bit.
 *
 * Only Scenario 4 (both beads 2-state, :424-519 / :457-606) is live in the
 * reference; other scenarios return an error here.
 */
#include "orc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EPSILONE 1.0e-6 /* UCG/pair_table_ucg_bethe.cpp:30 */

typedef struct {
  double u[2][2];  /* u[si][sj], already * factor_lj */
  double fp[2][2]; /* f/r       , already * factor_lj */
} quad;

/* the si/sj double loop of Scenario 4 (:425-505): four table evaluations */
static int eval4(const orc_pair *p, int itype, int jtype, double rsq, double factor_lj, quad *q)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  for (int si = 0; si < 2; si++) {
    int itype_si = p->formal_from_actual[itype * ms + si];
    for (int sj = 0; sj < 2; sj++) {
      int jtype_sj = p->formal_from_actual[jtype * ms + sj];
      const orc_table *tb = &p->tables[p->tabindex[itype_si * nt + jtype_sj]];
      double value, evdwl;
      int rc = orc_table_eval(tb, p->tabstyle, p->tablength, rsq, &value, &evdwl);
      if (rc) return rc;
      q->fp[si][sj] = factor_lj * value;
      evdwl *= factor_lj;
      q->u[si][sj] = evdwl;
    }
  }
  return 0;
}

static void ev_tally(orc_ev *ev, int eflag, int vflag, double evdwl, double fpair, double delx,
                     double dely, double delz, double scale)
{
  /* upstream Pair::ev_tally, global accumulators only; scale = 1
     for a pair seen once (newton on), 0.5 per owned end otherwise */
  if (eflag) ev->eng_vdwl += scale * evdwl;
  if (vflag) {
    ev->virial[0] += scale * (delx * delx * fpair);
    ev->virial[1] += scale * (dely * dely * fpair);
    ev->virial[2] += scale * (delz * delz * fpair);
    ev->virial[3] += scale * (delx * dely * fpair);
    ev->virial[4] += scale * (delx * delz * fpair);
    ev->virial[5] += scale * (dely * delz * fpair);
  }
}

/* Bethe closure, UCG/pair_table_ucg_bethe.cpp:544-581 */
static void bethe_pij(const orc_pair *p, const quad *q, double pi1, double pj1, double *pij00,
                      double *pij01, double *pij10, double *pij11)
{
  const double kT = p->kT;
  double Jij = q->u[1][1] + q->u[0][0] - q->u[0][1] - q->u[1][0];
  if (Jij / kT < -709.0) Jij = -700.0 * kT;
  double bij = orc_exp(-Jij / kT);
  double aij = orc_expm1(-Jij / kT);
  double Qij = (pi1 + pj1) * aij + 1.;
  double Dij = Qij * Qij - 4. * aij * bij * pi1 * pj1;
  Dij = (Dij > 0.0) ? Dij : 0.0; /* std::max(Dij, 0.0) */
  double p11 = 0.0;
  if (p->method_flag == ORC_METHOD_BETHE) {
    if (fabs(aij) < EPSILONE) {
      p11 = pi1 * pj1;
    } else {
      if (Qij < 0.0)
        p11 = (Qij - sqrt(Dij)) / (2. * aij);
      else
        p11 = (2. * bij * pi1 * pj1) / (Qij + sqrt(Dij));
    }
  } else {
    p11 = pi1 * pj1;
  }
  *pij11 = p11;
  *pij00 = 1. + p11 - pi1 - pj1;
  *pij10 = pi1 - p11;
  *pij01 = pj1 - p11;
}

/* prior of one bead the way the Bethe i-loop head does it (:179-205) */
static int bethe_prior_i(orc_pair *p, const orc_atoms *a, int i, double *p0, double *p1)
{
  const int ms = p->max_states;
  const int itype = a->type[i];
  if (a->ucgp[i] < -0.999) {
    if (p->prior_flag == ORC_PRIOR_CHEMPOT) {
      *p0 = p->prior_prob_from_type[itype * ms + 0];
      *p1 = p->prior_prob_from_type[itype * ms + 1];
    } else if (p->prior_flag == ORC_PRIOR_CHEMPOT_NOISE) {
      double randnum = (orc_ranmars_uniform(&p->random) - 0.5) * 2;
      randnum *= p->noise_level;
      double v = p->prior_prob_from_type[itype * ms + 0] + randnum;
      v = (v > 0.0) ? v : 0.0;
      *p0 = (0.999999 < v) ? 0.999999 : v;
      *p1 = 1. - *p0;
    } else {
      *p0 = 1.0 - a->ucgl[i];
      *p1 = a->ucgl[i];
    }
  } else {
    *p1 = a->ucgl[i];
    *p0 = 1.0 - a->ucgl[i];
  }
  return 0;
}

/* prior of a neighbour the way the Bethe j-loop does it (:227-253; the shipped
 * code indexes prior_prob_from_type with itype for the CHEMICAL_POTENTIAL route,
 * App. B #15 -- reproduced, it is moot with one actual type) */
static int bethe_prior_j(orc_pair *p, const orc_atoms *a, int itype, int j, double *p0, double *p1)
{
  const int ms = p->max_states;
  const int jtype = a->type[j];
  if (a->ucgp[j] < -0.999) {
    if (p->prior_flag == ORC_PRIOR_CHEMPOT) {
      *p0 = p->prior_prob_from_type[itype * ms + 0];
      *p1 = p->prior_prob_from_type[itype * ms + 1];
    } else if (p->prior_flag == ORC_PRIOR_CHEMPOT_NOISE) {
      double randnum = (orc_ranmars_uniform(&p->random) - 0.5) * 2;
      randnum *= p->noise_level;
      double v = p->prior_prob_from_type[jtype * ms + 0] + randnum;
      v = (v > 0.0) ? v : 0.0;
      *p0 = (0.999999 < v) ? 0.999999 : v;
      *p1 = 1. - *p0;
    } else {
      *p0 = 1.0 - a->ucgl[j];
      *p1 = a->ucgl[j];
    }
  } else {
    *p1 = a->ucgp[j];
    *p0 = 1.0 - a->ucgp[j];
  }
  return 0;
}

static int check_types(orc_pair *p, const orc_atoms *a, int nall)
{
  for (int i = 0; i < nall; i++) {
    int t = a->type[i];
    if (t < 1 || t > p->n_actual || p->n_states_per_type[t] != 2) {
      snprintf(p->errmsg, sizeof p->errmsg,
               "atom %d has type %d which is not a 2-state actual type: only Scenario 4 is live", i, t);
      return 1;
    }
  }
  return 0;
}

int orc_pair_compute_half(orc_pair *p, orc_atoms *a, const orc_list *l, int newton_pair,
                          int eflag, int vflag, orc_ev *ev)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  const int nlocal = a->nlocal;
  const int nall = a->nlocal + a->nghost;
  const double kT = p->kT;
  double *x = a->x, *f = a->f;
  const int evflag = eflag || vflag;
  memset(ev, 0, sizeof(*ev));
  if (p->style == ORC_STYLE_BETHE_DENSITY) {
    strcpy(p->errmsg, "use orc_pair_density_compute for table_ucg_bethe_density");
    return 1;
  }
  if (check_types(p, a, nall)) return 1;

  if (p->style == ORC_STYLE_UCGLD) {
    /* prologue :170-180 */
    for (int ii = 0; ii < l->inum; ii++) {
      int i = l->ilist[ii];
      int itype = a->type[i];
      a->num_ucgstates[i] = p->n_states_per_type[itype];
      if (p->n_states_per_type[itype] > 1) {
        double mui = p->chem_pot[p->formal_from_actual[itype * ms + 1]] -
            p->chem_pot[p->formal_from_actual[itype * ms + 0]];
        a->ucgforce[i] -= mui;
        a->scores[2 * i + 1] -= mui / kT;
      }
    }
  } else {
    /* UCG/pair_table_ucg_bethe.cpp:155-162: scores are ASSIGNED */
    for (int ii = 0; ii < l->inum; ii++) {
      int i = l->ilist[ii];
      int itype = a->type[i];
      a->num_ucgstates[i] = p->n_states_per_type[itype];
      for (int si = 0; si < p->n_states_per_type[itype]; si++)
        a->scores[2 * i + si] = -p->chem_pot[p->formal_from_actual[itype * ms + si]] / kT;
    }
  }

  for (int ii = 0; ii < l->inum; ii++) {
    const int i = l->ilist[ii];
    const int itype = a->type[i];
    const int *jlist = l->neigh + l->first[ii];
    const int jnum = l->numneigh[ii];
    const int istate = a->ucgstate[i];
    const double ldi = a->ucgl[i];
    const double xtmp = x[3 * i + 0], ytmp = x[3 * i + 1], ztmp = x[3 * i + 2];
    double pi0 = 0.0, pi1 = 0.0;
    if (p->style == ORC_STYLE_BETHE) bethe_prior_i(p, a, i, &pi0, &pi1);

    for (int jj = 0; jj < jnum; jj++) {
      int j = jlist[jj];
      const double factor_lj = p->special_lj[(j >> ORC_SBBITS) & 3];
      j &= ORC_NEIGHMASK;
      const int jtype = a->type[j];
      const int jstate = a->ucgstate[j];
      const double ldj = a->ucgl[j];
      const double delx = xtmp - x[3 * j + 0];
      const double dely = ytmp - x[3 * j + 1];
      const double delz = ztmp - x[3 * j + 2];
      const double rsq = delx * delx + dely * dely + delz * delz;
      double pj0 = 0.0, pj1 = 0.0;
      if (p->style == ORC_STYLE_BETHE) bethe_prior_j(p, a, itype, j, &pj0, &pj1);

      if (rsq < p->cutsq[itype * nt + jtype]) {
        quad q;
        int rc = eval4(p, itype, jtype, rsq, factor_lj, &q);
        if (rc) {
          if (!ev->err) { ev->err = rc; ev->err_i = i; ev->err_j = j; }
          continue;
        }
        double evdwl, fpair;
        const int jok = (j < nlocal || newton_pair);
        if (p->style == ORC_STYLE_UCGLD || p->pseudo_flag == 0) {
          /* pseudo-likelihood scores :492-502 / bethe :526-539 */
          for (int si = 0; si < 2; si++)
            for (int sj = 0; sj < 2; sj++) {
              if (sj == jstate) a->scores[2 * i + si] -= q.u[si][sj] / kT;
              if (si == istate && jok) a->scores[2 * j + sj] -= q.u[si][sj] / kT;
            }
        }
        const double u00 = q.u[0][0], u01 = q.u[0][1], u10 = q.u[1][0], u11 = q.u[1][1];
        const double fpair00 = q.fp[0][0], fpair01 = q.fp[0][1], fpair10 = q.fp[1][0], fpair11 = q.fp[1][1];
        if (p->style == ORC_STYLE_UCGLD) {
          /* :507-517 */
          evdwl = (1. - ldi) * (1. - ldj) * u00 + (1. - ldi) * ldj * u01 + (1. - ldj) * ldi * u10 + ldi * ldj * u11;
          fpair = (1. - ldi) * (1. - ldj) * fpair00 + (1. - ldi) * ldj * fpair01 + (1. - ldj) * ldi * fpair10 + ldi * ldj * fpair11;
          a->ucgforce[i] -= ldj * (u11 - u01) + (1. - ldj) * (u10 - u00);
          if (jok) a->ucgforce[j] -= ldi * (u11 - u10) + (1. - ldi) * (u01 - u00);
        } else {
          double pij00, pij01, pij10, pij11;
          bethe_pij(p, &q, pi1, pj1, &pij00, &pij01, &pij10, &pij11);
          if (p->pseudo_flag == 1) {
            /* full SCE scores exactly as shipped, :583-601 */
            double pj0i0 = pij00 / pi0, pj0i1 = pij01 / pi0, pj1i0 = pij10 / pi1, pj1i1 = pij11 / pi1;
            double pi0j0 = pij00 / pj0, pi0j1 = pij10 / pj0, pi1j0 = pij01 / pj1, pi1j1 = pij11 / pj1;
            a->scores[2 * i + 0] -= (pj0i0 * u00 + pj1i0 * u01) / kT;
            a->scores[2 * i + 1] -= (pj0i1 * u10 + pj1i1 * u11) / kT;
            if (jok) {
              a->scores[2 * j + 0] -= (pi0j0 * u00 + pi0j1 * u01) / kT;
              a->scores[2 * j + 1] -= (pi1j0 * u10 + pi1j1 * u11) / kT;
            }
          }
          evdwl = pij00 * u00 + pij01 * u01 + pij10 * u10 + pij11 * u11;
          fpair = pij00 * fpair00 + pij01 * fpair01 + pij10 * fpair10 + pij11 * fpair11;
        }
        f[3 * i + 0] += delx * fpair;
        f[3 * i + 1] += dely * fpair;
        f[3 * i + 2] += delz * fpair;
        if (jok) {
          f[3 * j + 0] -= delx * fpair;
          f[3 * j + 1] -= dely * fpair;
          f[3 * j + 2] -= delz * fpair;
        }
        if (evflag) ev_tally(ev, eflag, vflag, evdwl, fpair, delx, dely, delz, 1.0);
      }
    }
  }
  return ev->err ? 2 : 0;
}

/* pair_once: image of one term in units of 2^-40 (round to nearest even, as the double addition does) */
static long long once_image(const orc_pair *p, double v, orc_ev *ev)
{
  union { double d; long long i; } t, m;
  if (!(fabs(v) < p->once_limit)) ev->err |= 4;
  t.d = v + ORC_ONCE_MAGIC;
  m.d = ORC_ONCE_MAGIC;
  return t.i - m.i;
}

int orc_pair_compute_gather(orc_pair *p, orc_atoms *a, const orc_list *l, int eflag, int vflag,
                            orc_ev *ev)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  const int nall = a->nlocal + a->nghost;
  const double kT = p->kT;
  const double *x = a->x;
  const int B = p->once_block;
  long long *acc = NULL; /* pair_once: six integer accumulators per owned bead */
  memset(ev, 0, sizeof(*ev));
  if (B > 0) {
    if (p->style != ORC_STYLE_UCGLD) {
      strcpy(p->errmsg, "pair_once order is defined for table_ucgld");
      return 1;
    }
    acc = (long long *) calloc((size_t) a->nlocal * 6 + 1, sizeof(long long));
  }
  if (p->style == ORC_STYLE_BETHE_DENSITY) {
    strcpy(p->errmsg, "use orc_pair_density_compute for table_ucg_bethe_density");
    return 1;
  }
  if (p->style == ORC_STYLE_BETHE && p->prior_flag == ORC_PRIOR_CHEMPOT_NOISE) {
    strcpy(p->errmsg, "prior chemical_potential noise draws RNG in list order (App. B #17): half-list mode only");
    return 1;
  }
  if (check_types(p, a, nall)) return 1;

  for (int ii = 0; ii < l->inum; ii++) {
    const int k = l->ilist[ii];
    const int ktype = a->type[k];
    const int kstate = a->ucgstate[k];
    const double lk = a->ucgl[k];
    const int *row = l->neigh + l->first[ii];
    const int n = l->numneigh[ii];
    const double xk = x[3 * k + 0], yk = x[3 * k + 1], zk = x[3 * k + 2];
    /* slot accumulators: entry e of the row goes to slot e % S; slot 0 starts from the prologue
       value, the others from 0; at the end slots are combined by the fixed tree
       s[l] += s[l + S/2], ..., s[l] += s[l + 1] (what a group of S GPU lanes does) */
    const int S = p->gather_slots > 0 ? p->gather_slots : 1;
    double sfx[ORC_MAX_SLOTS] = {0}, sfy[ORC_MAX_SLOTS] = {0}, sfz[ORC_MAX_SLOTS] = {0};
    double suf[ORC_MAX_SLOTS] = {0}, ss0[ORC_MAX_SLOTS] = {0}, ss1[ORC_MAX_SLOTS] = {0};
    double fx = 0.0, fy = 0.0, fz = 0.0, uf = 0.0, s0 = 0.0, s1 = 0.0;
    double e_acc = 0.0, v_acc[6] = {0, 0, 0, 0, 0, 0};

    a->num_ucgstates[k] = p->n_states_per_type[ktype];
    if (p->style == ORC_STYLE_UCGLD) {
      double mui = p->chem_pot[p->formal_from_actual[ktype * ms + 1]] -
          p->chem_pot[p->formal_from_actual[ktype * ms + 0]];
      uf -= mui;
      s1 -= mui / kT;
    } else {
      s0 = -p->chem_pot[p->formal_from_actual[ktype * ms + 0]] / kT;
      s1 = -p->chem_pot[p->formal_from_actual[ktype * ms + 1]] / kT;
    }

    sfx[0] = fx; sfy[0] = fy; sfz[0] = fz; suf[0] = uf; ss0[0] = s0; ss1[0] = s1;
    int ekept = 0; /* position in the row as the library stores it (pair_once rows lack the dropped entries) */
    for (int e = 0; e < n; e++) {
      int own_pair = 0;
      if (B > 0) {
        const int mm = row[e] & ORC_NEIGHMASK;
        if (mm < a->nlocal && k / B == mm / B) {
          const int k_keeps = (k < mm) != (((k + mm) & 1) != 0);
          if (!k_keeps) continue; /* swept from the partner's row */
          own_pair = 1;
        }
      }
      const int slot = ekept % S;
      ekept++;
      fx = sfx[slot]; fy = sfy[slot]; fz = sfz[slot]; uf = suf[slot]; s0 = ss0[slot]; s1 = ss1[slot];
      int m = row[e];
      const int k_is_i = (m >> ORC_ORIENT_BIT) & 1;
      const double factor_lj = p->special_lj[(m >> ORC_SBBITS) & 3];
      m &= ORC_NEIGHMASK;
      const int mtype = a->type[m];
      const int mstate = a->ucgstate[m];
      const double lm = a->ucgl[m];
      /* the reference's orientation of this pair */
      const int i = k_is_i ? k : m, j = k_is_i ? m : k;
      const int itype = k_is_i ? ktype : mtype, jtype = k_is_i ? mtype : ktype;
      const double delx = x[3 * i + 0] - x[3 * j + 0];
      const double dely = x[3 * i + 1] - x[3 * j + 1];
      const double delz = x[3 * i + 2] - x[3 * j + 2];
      const double rsq = delx * delx + dely * dely + delz * delz;
      (void) xk; (void) yk; (void) zk;
      if (rsq < p->cutsq[itype * nt + jtype]) {
        quad q;
        int rc = eval4(p, itype, jtype, rsq, factor_lj, &q);
        if (rc) {
          if (!ev->err) { ev->err = rc; ev->err_i = k; ev->err_j = m; }
          continue;
        }
        const int istate = k_is_i ? kstate : mstate, jstate = k_is_i ? mstate : kstate;
        const double ldi = k_is_i ? lk : lm, ldj = k_is_i ? lm : lk;
        const double u00 = q.u[0][0], u01 = q.u[0][1], u10 = q.u[1][0], u11 = q.u[1][1];
        const double fpair00 = q.fp[0][0], fpair01 = q.fp[0][1], fpair10 = q.fp[1][0], fpair11 = q.fp[1][1];
        double evdwl, fpair;
        if (p->style == ORC_STYLE_UCGLD || p->pseudo_flag == 0) {
          if (k_is_i) {
            s0 -= q.u[0][jstate] / kT;
            s1 -= q.u[1][jstate] / kT;
          } else {
            s0 -= q.u[istate][0] / kT;
            s1 -= q.u[istate][1] / kT;
          }
        }
        if (p->style == ORC_STYLE_UCGLD) {
          evdwl = (1. - ldi) * (1. - ldj) * u00 + (1. - ldi) * ldj * u01 + (1. - ldj) * ldi * u10 + ldi * ldj * u11;
          fpair = (1. - ldi) * (1. - ldj) * fpair00 + (1. - ldi) * ldj * fpair01 + (1. - ldj) * ldi * fpair10 + ldi * ldj * fpair11;
          if (k_is_i)
            uf -= ldj * (u11 - u01) + (1. - ldj) * (u10 - u00);
          else
            uf -= ldi * (u11 - u10) + (1. - ldi) * (u01 - u00);
        } else {
          /* priors: i from ucgl[i], j from ucgp[j] (first call: per prior_flag) */
          double pi0, pi1, pj0, pj1;
          bethe_prior_i(p, a, i, &pi0, &pi1);
          bethe_prior_j(p, a, itype, j, &pj0, &pj1);
          double pij00, pij01, pij10, pij11;
          bethe_pij(p, &q, pi1, pj1, &pij00, &pij01, &pij10, &pij11);
          if (p->pseudo_flag == 1) {
            double pj0i0 = pij00 / pi0, pj0i1 = pij01 / pi0, pj1i0 = pij10 / pi1, pj1i1 = pij11 / pi1;
            double pi0j0 = pij00 / pj0, pi0j1 = pij10 / pj0, pi1j0 = pij01 / pj1, pi1j1 = pij11 / pj1;
            if (k_is_i) {
              s0 -= (pj0i0 * u00 + pj1i0 * u01) / kT;
              s1 -= (pj0i1 * u10 + pj1i1 * u11) / kT;
            } else {
              s0 -= (pi0j0 * u00 + pi0j1 * u01) / kT;
              s1 -= (pi1j0 * u10 + pi1j1 * u11) / kT;
            }
          }
          evdwl = pij00 * u00 + pij01 * u01 + pij10 * u10 + pij11 * u11;
          fpair = pij00 * fpair00 + pij01 * fpair01 + pij10 * fpair10 + pij11 * fpair11;
        }
        if (k_is_i) {
          fx += delx * fpair;
          fy += dely * fpair;
          fz += delz * fpair;
        } else {
          fx -= delx * fpair;
          fy -= dely * fpair;
          fz -= delz * fpair;
        }
        /* a pair is seen from both of its owned ends: half of E and W each time */
        if (eflag) e_acc += 0.5 * evdwl;
        if (vflag) {
          v_acc[0] += 0.5 * (delx * delx * fpair);
          v_acc[1] += 0.5 * (dely * dely * fpair);
          v_acc[2] += 0.5 * (delz * delz * fpair);
          v_acc[3] += 0.5 * (delx * dely * fpair);
          v_acc[4] += 0.5 * (delx * delz * fpair);
          v_acc[5] += 0.5 * (dely * delz * fpair);
        }
        if (own_pair) {
          /* pair_once: this row is the only one holding the pair; what the half-list sweep adds to the partner m
             (UCG/pair_table_ucgld.cpp:500-502 / :492-498, :514-517, :523-530) goes to m's integer accumulators */
          long long *am = acc + (size_t) m * 6;
          double ufm, sm0, sm1;
          if (k_is_i) { /* m is "j" */
            am[0] += once_image(p, -(delx * fpair), ev);
            am[1] += once_image(p, -(dely * fpair), ev);
            am[2] += once_image(p, -(delz * fpair), ev);
            ufm = ldi * (u11 - u10) + (1. - ldi) * (u01 - u00);
            sm0 = q.u[istate][0] / kT;
            sm1 = q.u[istate][1] / kT;
          } else { /* m is "i" */
            am[0] += once_image(p, delx * fpair, ev);
            am[1] += once_image(p, dely * fpair, ev);
            am[2] += once_image(p, delz * fpair, ev);
            ufm = ldj * (u11 - u01) + (1. - ldj) * (u10 - u00);
            sm0 = q.u[0][jstate] / kT;
            sm1 = q.u[1][jstate] / kT;
          }
          am[3] += once_image(p, -ufm, ev);
          am[4] += once_image(p, -sm0, ev);
          am[5] += once_image(p, -sm1, ev);
          if (eflag) e_acc += 0.5 * evdwl;
          if (vflag) {
            v_acc[0] += 0.5 * (delx * delx * fpair);
            v_acc[1] += 0.5 * (dely * dely * fpair);
            v_acc[2] += 0.5 * (delz * delz * fpair);
            v_acc[3] += 0.5 * (delx * dely * fpair);
            v_acc[4] += 0.5 * (delx * delz * fpair);
            v_acc[5] += 0.5 * (dely * delz * fpair);
          }
        }
      }
      sfx[slot] = fx; sfy[slot] = fy; sfz[slot] = fz; suf[slot] = uf; ss0[slot] = s0; ss1[slot] = s1;
    }
    for (int off = S / 2; off > 0; off >>= 1)
      for (int l = 0; l < off; l++) {
        sfx[l] += sfx[l + off]; sfy[l] += sfy[l + off]; sfz[l] += sfz[l + off];
        suf[l] += suf[l + off]; ss0[l] += ss0[l + off]; ss1[l] += ss1[l + off];
      }
    fx = sfx[0]; fy = sfy[0]; fz = sfz[0]; uf = suf[0]; s0 = ss0[0]; s1 = ss1[0];
    a->f[3 * k + 0] = fx;
    a->f[3 * k + 1] = fy;
    a->f[3 * k + 2] = fz;
    if (p->style == ORC_STYLE_UCGLD) a->ucgforce[k] = uf;
    a->scores[2 * k + 0] = s0;
    a->scores[2 * k + 1] = s1;
    ev->eng_vdwl += e_acc;
    for (int c = 0; c < 6; c++) ev->virial[c] += v_acc[c];
  }
  if (acc) {
    for (int ii = 0; ii < l->inum; ii++) {
      const int k = l->ilist[ii];
      const long long *ak = acc + (size_t) k * 6;
      a->f[3 * k + 0] += (double) ak[0] * ORC_ONCE_UNIT;
      a->f[3 * k + 1] += (double) ak[1] * ORC_ONCE_UNIT;
      a->f[3 * k + 2] += (double) ak[2] * ORC_ONCE_UNIT;
      a->ucgforce[k] += (double) ak[3] * ORC_ONCE_UNIT;
      a->scores[2 * k + 0] += (double) ak[4] * ORC_ONCE_UNIT;
      a->scores[2 * k + 1] += (double) ak[5] * ORC_ONCE_UNIT;
    }
    free(acc);
  }
  return ev->err ? 2 : 0;
}
