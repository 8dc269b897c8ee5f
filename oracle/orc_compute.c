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

/* ---- "fixed" sums: the integer image of one term: round to nearest even of v * 2^(38 + e); field = 0 force
   components, 1 ucgforce (energies), 2 scores (energies / kT); |image| < 2^62 or error bit 4 */
static long long sum_image(const orc_pair *p, int field, double v, orc_ev *ev)
{
  const double scaled = ldexp(v, 38 + p->sum_exp[field]); /* exact: a power of two */
  if (!(fabs(scaled) < 4.611686018427388e18)) {
    ev->err |= 4;
    return 0;
  }
  return llrint(scaled);
}

static double sum_decode(const orc_pair *p, int field, long long s)
{
  return ldexp((double) s, -38 - p->sum_exp[field]);
}

/* The units of the integer sums: powers of two fixed by the tables (so that they follow the unit system): with
   Fref = max |f(k)| sqrt(rsq_k) and Uref = max |e(k)| over the reachable tables and the knots whose r^2 lies in the
   upper three quarters of the table's r^2 range (the well-behaved part of the grid), the exponents are
   4 - ilogb(Fref), 4 - ilogb(Uref), 4 - ilogb(Uref / kT): the reference magnitudes map to [16, 32).  Zero or
   non-finite references give exponent 0.  (Not defined for BITMAP tables: the library's kernels that sum this way do
   not take them.) */
void orc_pair_sum_scales(orc_pair *p)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  double fref = 0.0, uref = 0.0;
  p->sum_exp[0] = p->sum_exp[1] = p->sum_exp[2] = 0;
  if (p->tabstyle == ORC_BITMAP) return;
  for (int ti = 1; ti <= p->n_actual; ti++)
    for (int tj = 1; tj <= p->n_actual; tj++)
      for (int sa = 0; sa < 2; sa++)
        for (int sb = 0; sb < 2; sb++) {
          const int fi = p->formal_from_actual[ti * ms + sa], fj = p->formal_from_actual[tj * ms + sb];
          if (fi < 1 || fj < 1 || fi > p->n_formal || fj > p->n_formal) continue;
          const orc_table *tb = &p->tables[p->tabindex[fi * nt + fj]];
          const int n = (p->tabstyle == ORC_LOOKUP) ? p->tablength - 1 : p->tablength;
          const double from = tb->innersq + 0.25 * ((n - 1) * tb->delta);
          for (int k = 0; k < n; k++) {
            const double rsq = tb->innersq + k * tb->delta;
            if (!(rsq >= from)) continue;
            const double fv = fabs(tb->f[k]) * sqrt(rsq > 0.0 ? rsq : 0.0), ue = fabs(tb->e[k]);
            if (fv > fref) fref = fv;
            if (ue > uref) uref = ue;
          }
        }
  const double sref = uref / p->kT;
  p->sum_exp[0] = (fref > 0.0 && isfinite(fref)) ? 4 - ilogb(fref) : 0;
  p->sum_exp[1] = (uref > 0.0 && isfinite(uref)) ? 4 - ilogb(uref) : 0;
  p->sum_exp[2] = (sref > 0.0 && isfinite(sref)) ? 4 - ilogb(sref) : 0;
}

void orc_pair_set_sum_fixed(orc_pair *p, int on) { p->sum_fixed = on ? 1 : 0; }

/* one running sum of a bead: ordered mode adds doubles in row order; fixed mode keeps the prologue value in d and
   adds the terms' integer images to i */
typedef struct { double d; long long i; } acc_t;

static inline void acc_add(const orc_pair *p, int field, acc_t *a, double v, orc_ev *ev)
{
  if (p->sum_fixed) a->i += sum_image(p, field, v, ev);
  else a->d += v;
}

int orc_pair_compute_gather(orc_pair *p, orc_atoms *a, const orc_list *l, int eflag, int vflag,
                            orc_ev *ev)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  const int nall = a->nlocal + a->nghost;
  const double kT = p->kT;
  const double *x = a->x;
  memset(ev, 0, sizeof(*ev));
  if (p->style == ORC_STYLE_BETHE_DENSITY) {
    strcpy(p->errmsg, "use orc_pair_density_compute for table_ucg_bethe_density");
    return 1;
  }
  if (p->style == ORC_STYLE_BETHE && p->prior_flag == ORC_PRIOR_CHEMPOT_NOISE) {
    strcpy(p->errmsg, "prior chemical_potential noise draws RNG in list order (App. B #17): half-list mode only");
    return 1;
  }
  if (p->sum_fixed && p->tabstyle == ORC_BITMAP) {
    strcpy(p->errmsg, "fixed sums are not defined for BITMAP tables");
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
    /* ordered mode -- slot accumulators: entry e of the row goes to slot e % S; slot 0 starts from the prologue
       value, the others from 0; at the end slots are combined by the fixed tree
       s[l] += s[l + S/2], ..., s[l] += s[l + 1] (what a group of S GPU lanes does).
       fixed mode -- one slot: the integer sum does not depend on any order */
    const int S = (!p->sum_fixed && p->gather_slots > 0) ? p->gather_slots : 1;
    enum { FX, FY, FZ, UF, S0, S1 };
    static const int field_of[6] = {0, 0, 0, 1, 2, 2};
    acc_t acc[6][ORC_MAX_SLOTS];
    memset(acc, 0, sizeof(acc));
    double e_acc = 0.0, v_acc[6] = {0, 0, 0, 0, 0, 0};

    a->num_ucgstates[k] = p->n_states_per_type[ktype];
    if (p->style == ORC_STYLE_UCGLD) {
      double mui = p->chem_pot[p->formal_from_actual[ktype * ms + 1]] -
          p->chem_pot[p->formal_from_actual[ktype * ms + 0]];
      acc[UF][0].d -= mui;
      acc[S1][0].d -= mui / kT;
    } else {
      acc[S0][0].d = -p->chem_pot[p->formal_from_actual[ktype * ms + 0]] / kT;
      acc[S1][0].d = -p->chem_pot[p->formal_from_actual[ktype * ms + 1]] / kT;
    }

    for (int e = 0; e < n; e++) {
      const int slot = e % S;
      acc_t *fx = &acc[FX][slot], *fy = &acc[FY][slot], *fz = &acc[FZ][slot];
      acc_t *uf = &acc[UF][slot], *s0 = &acc[S0][slot], *s1 = &acc[S1][slot];
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
        /* (x -= y and x += -y are the same double operation) */
        if (p->style == ORC_STYLE_UCGLD || p->pseudo_flag == 0) {
          if (k_is_i) {
            acc_add(p, 2, s0, -(q.u[0][jstate] / kT), ev);
            acc_add(p, 2, s1, -(q.u[1][jstate] / kT), ev);
          } else {
            acc_add(p, 2, s0, -(q.u[istate][0] / kT), ev);
            acc_add(p, 2, s1, -(q.u[istate][1] / kT), ev);
          }
        }
        if (p->style == ORC_STYLE_UCGLD) {
          evdwl = (1. - ldi) * (1. - ldj) * u00 + (1. - ldi) * ldj * u01 + (1. - ldj) * ldi * u10 + ldi * ldj * u11;
          fpair = (1. - ldi) * (1. - ldj) * fpair00 + (1. - ldi) * ldj * fpair01 + (1. - ldj) * ldi * fpair10 + ldi * ldj * fpair11;
          if (k_is_i)
            acc_add(p, 1, uf, -(ldj * (u11 - u01) + (1. - ldj) * (u10 - u00)), ev);
          else
            acc_add(p, 1, uf, -(ldi * (u11 - u10) + (1. - ldi) * (u01 - u00)), ev);
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
              acc_add(p, 2, s0, -((pj0i0 * u00 + pj1i0 * u01) / kT), ev);
              acc_add(p, 2, s1, -((pj0i1 * u10 + pj1i1 * u11) / kT), ev);
            } else {
              acc_add(p, 2, s0, -((pi0j0 * u00 + pi0j1 * u01) / kT), ev);
              acc_add(p, 2, s1, -((pi1j0 * u10 + pi1j1 * u11) / kT), ev);
            }
          }
          evdwl = pij00 * u00 + pij01 * u01 + pij10 * u10 + pij11 * u11;
          fpair = pij00 * fpair00 + pij01 * fpair01 + pij10 * fpair10 + pij11 * fpair11;
        }
        if (k_is_i) {
          acc_add(p, 0, fx, delx * fpair, ev);
          acc_add(p, 0, fy, dely * fpair, ev);
          acc_add(p, 0, fz, delz * fpair, ev);
        } else {
          acc_add(p, 0, fx, -(delx * fpair), ev);
          acc_add(p, 0, fy, -(dely * fpair), ev);
          acc_add(p, 0, fz, -(delz * fpair), ev);
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
      }
    }
    double tot[6];
    for (int c = 0; c < 6; c++) {
      if (p->sum_fixed) {
        tot[c] = acc[c][0].d + sum_decode(p, field_of[c], acc[c][0].i);
      } else {
        for (int off = S / 2; off > 0; off >>= 1)
          for (int s = 0; s < off; s++) acc[c][s].d += acc[c][s + off].d;
        tot[c] = acc[c][0].d;
      }
    }
    a->f[3 * k + 0] = tot[FX];
    a->f[3 * k + 1] = tot[FY];
    a->f[3 * k + 2] = tot[FZ];
    if (p->style == ORC_STYLE_UCGLD) a->ucgforce[k] = tot[UF];
    a->scores[2 * k + 0] = tot[S0];
    a->scores[2 * k + 1] = tot[S1];
    ev->eng_vdwl += e_acc;
    for (int c = 0; c < 6; c++) ev->virial[c] += v_acc[c];
  }
  return ev->err ? 2 : 0;
}
