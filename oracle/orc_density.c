/* orc_density.c -- oracle compute() for table_ucg_bethe_density + small accessors.
 * TEST INFRASTRUCTURE (see orc.h). */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>


/* ---- accessors used by the tests to compare the product's tables bit for bit ---- */

void orc_pair_table_info(const orc_pair *p, int m, double *out)
{
  const orc_table *tb = &p->tables[m];
  out[0] = tb->innersq;
  out[1] = tb->delta;
  out[2] = tb->invdelta;
  out[3] = tb->deltasq6;
  out[4] = tb->cut;
  out[5] = tb->ninput;
  out[6] = tb->match;
  out[7] = p->ntables;
}

const double *orc_pair_table_array(const orc_pair *p, int m, const char *name, int *n)
{
  const orc_table *tb = &p->tables[m];
  const int tl = p->tablength, tlm1 = tl - 1;
  const int lookup = (p->tabstyle == ORC_LOOKUP);
  *n = 0;
  if (p->tabstyle == ORC_BITMAP) { /* every array of a bitmapped table is 2^tablength long */
    const int nt = 1 << tl;
    const double *v = !strcmp(name, "rsq") ? tb->rsq : !strcmp(name, "e") ? tb->e : !strcmp(name, "f") ? tb->f :
                      !strcmp(name, "de") ? tb->de : !strcmp(name, "df") ? tb->df : !strcmp(name, "drsq") ? tb->drsq : NULL;
    if (v) { *n = nt; return v; }
    if (!strcmp(name, "bits")) { /* nmask, nshiftbits as doubles */
      static double bits[2];
      bits[0] = tb->nmask; bits[1] = tb->nshiftbits; *n = 2; return bits;
    }
  }
  if (!strcmp(name, "rfile")) { *n = tb->ninput; return tb->rfile; }
  if (!strcmp(name, "efile")) { *n = tb->ninput; return tb->efile; }
  if (!strcmp(name, "ffile")) { *n = tb->ninput; return tb->ffile; }
  if (!strcmp(name, "e2file")) { *n = tb->e2file ? tb->ninput : 0; return tb->e2file; }
  if (!strcmp(name, "f2file")) { *n = tb->f2file ? tb->ninput : 0; return tb->f2file; }
  if (!strcmp(name, "rsq")) { *n = tb->rsq ? tl : 0; return tb->rsq; }
  if (!strcmp(name, "e")) { *n = lookup ? tlm1 : tl; return tb->e; }
  if (!strcmp(name, "f")) { *n = lookup ? tlm1 : tl; return tb->f; }
  if (!strcmp(name, "de")) { *n = tb->de ? tlm1 : 0; return tb->de; }
  if (!strcmp(name, "df")) { *n = tb->df ? tlm1 : 0; return tb->df; }
  if (!strcmp(name, "e2")) { *n = tb->e2 ? tl : 0; return tb->e2; }
  if (!strcmp(name, "f2")) { *n = tb->f2 ? tl : 0; return tb->f2; }
  return NULL;
}

const int *orc_pair_int_array(const orc_pair *p, const char *name, int *n)
{
  const int nt = p->n_formal + 1;
  *n = 0;
  if (!strcmp(name, "tabindex")) { *n = nt * nt; return p->tabindex; }
  if (!strcmp(name, "setflag")) { *n = nt * nt; return p->setflag; }
  if (!strcmp(name, "n_states_per_type")) { *n = p->n_actual + 1; return p->n_states_per_type; }
  if (!strcmp(name, "formal_from_actual")) { *n = (p->n_actual + 1) * p->max_states; return p->formal_from_actual; }
  if (!strcmp(name, "actual_from_formal")) { *n = p->n_formal + 1; return p->actual_from_formal; }
  if (!strcmp(name, "use_density")) { *n = p->n_actual + 1; return p->use_density; }
  if (!strcmp(name, "use_state_entropy")) { *n = p->n_actual + 1; return p->use_state_entropy; }
  return NULL;
}

const double *orc_pair_dbl_array(const orc_pair *p, const char *name, int *n)
{
  const int nt = p->n_formal + 1;
  *n = 0;
  if (!strcmp(name, "cutsq")) { *n = nt * nt; return p->cutsq; }
  if (!strcmp(name, "chem_pot")) { *n = p->n_formal + 1; return p->chem_pot; }
  if (!strcmp(name, "prior_prob_from_type")) { *n = (p->n_actual + 1) * p->max_states; return p->prior_prob_from_type; }
  if (!strcmp(name, "cv_thresholds")) { *n = p->n_actual + 1; return p->cv_thresholds; }
  if (!strcmp(name, "threshold_radii")) { *n = p->n_actual + 1; return p->threshold_radii; }
  return NULL;
}

void orc_pair_set_special_lj(orc_pair *p, const double *s)
{
  for (int i = 0; i < 4; i++) p->special_lj[i] = s[i];
}

void orc_pair_set_gather_slots(orc_pair *p, int slots) { p->gather_slots = slots; }

void orc_pair_set_compat(orc_pair *p, int flags) { p->density_proximity_as_shipped = flags & 1; }

void orc_fix_langevin_get(const orc_fix_langevin *fx, double *out)
{
  out[0] = fx->t_target;
  out[1] = fx->tsqrt;
  out[2] = fx->lambda_temp;
  for (int i = 1; i <= fx->ntypes && i < 8; i++) {
    out[2 + 2 * i - 1] = fx->gfactor1[i];
    out[2 + 2 * i] = fx->gfactor2[i];
  }
}

orc_fix_ucgstate *orc_fix_ucgstate_create(int ld_flag, int mc_flag, int mc_seed, double mc_rate, int me)
{
  orc_fix_ucgstate *fx = (orc_fix_ucgstate *) calloc(1, sizeof(orc_fix_ucgstate));
  orc_fix_ucgstate_init(fx, ld_flag, mc_flag, mc_seed, mc_rate, me);
  return fx;
}

void orc_fix_ucgstate_destroy(orc_fix_ucgstate *fx) { free(fx); }

/* ------------------------------------------------------------------------------------------
 * table_ucg_bethe_density: PairTable_UCG_Bethe_Density::compute,
 * UCG/pair_table_ucg_bethe_density.cpp:133-758 (helpers :107-127), Scenario 4 (:529-658).
 * FULL list, newton off.  Three passes:
 *   1 (:219-274)  local density rho_i = sum_j w(r_ij) -> prior p_i0 = 1/2 + 1/2 tanh((rho-rho_th)/(0.1 rho_th))
 *   2 (:284-664)  tables, scores, Bethe closure, pair forces (x 1/2 when j is owned), entropic accumulators G
 *   3 (:669-734)  posterior -> ucgp, back-force of the density CV over the neighbours
 * Decisions where the shipped text is inconsistent or undefined, all flagged:
 *   #7  ghost priors: the reference's forward_comm moves 0 bytes (ghost priors = 0 -> NaN); here ghosts
 *       carry their owner's prior and CV force (what the legacy style does, rleucg_interface.cpp:131-160)
 *   #8  ucgp uses n_states_per_type[itype] (the shipped [i] indexes by atom)
 *   #9  closure kept as shipped: a = b - 1, D = sqrt(Q^2 - 4ab pi pj), p11 = (Q - D)/2/a, unguarded
 *   #11 neighbour indices are masked in all passes
 *   #12 back-force uses the DERIVATIVE of the proximity function (legacy :480); compat flag
 *       density_proximity_as_shipped = 1 uses the function itself as shipped (:719)
 *   back-force on / from ghost neighbours: as shipped a ghost j receives nothing and nothing comes
 *       back from it (momentum is lost across boundaries); here every neighbour m, owned or ghost,
 *       pushes back on k with its own CV force (legacy reverse_comm design, :104-129,:494)
 * mode 0 = sequential sweep with scatter (the reference's loop shape), mode 1 = canonical gather.
 */
#include <stdio.h>

static double prox(const orc_pair *p, int type, double r)
{
  double t = orc_tanh((r - p->threshold_radii[type]) / (0.1 * p->threshold_radii[type]));
  return 0.5 * (1.0 - t);
}

static double prox_der(const orc_pair *p, int type, double r)
{
  double t = orc_tanh((r - p->threshold_radii[type]) / (0.1 * p->threshold_radii[type]));
  return 0.5 * (1.0 - t * t) / (0.1 * p->threshold_radii[type]);
}

typedef struct { double u[2][2], fp[2][2]; } dquad;

static int deval4(const orc_pair *p, int itype, int jtype, double rsq, double factor_lj, dquad *q)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  for (int si = 0; si < 2; si++) {
    int fi = p->formal_from_actual[itype * ms + si];
    for (int sj = 0; sj < 2; sj++) {
      int fj = p->formal_from_actual[jtype * ms + sj];
      const orc_table *tb = &p->tables[p->tabindex[fi * nt + fj]];
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

/* closure as shipped (:608-622) for the pair seen with "i" holding prior pi1 */
static void dclosure(const orc_pair *p, const dquad *q, double pi1, double pj1, double *p00, double *p01,
                     double *p10, double *p11)
{
  const double kT = p->kT;
  double Jij = q->u[1][1] + q->u[0][0] - q->u[0][1] - q->u[1][0];
  double bij = orc_exp(-Jij / kT);
  double aij = bij - 1.;
  double Qij = (pi1 + pj1) * aij + 1.;
  double Dij = sqrt(Qij * Qij - 4. * aij * bij * pi1 * pj1);
  *p11 = (Qij - Dij) / 2. / aij;
  *p00 = 1. + *p11 - pi1 - pj1;
  *p10 = pi1 - *p11;
  *p01 = pj1 - *p11;
}

/* The three passes as separate calls on shared work arrays, so that a decomposed run (orc_md.c: orc_world) can move the
 * ghosts' priors (after pass 1) and CV forces (after pass 2) between ranks -- the forward_comm the reference declares and
 * never performs (UCG/pair_table_ucg_bethe_density.cpp:280, App. B #7).  orc_pair_density_compute below is the single-rank
 * composition: pass 1, ghosts <- owners, pass 2, ghosts <- owners, pass 3. */
orc_density_work *orc_density_work_create(int nall)
{
  orc_density_work *w = (orc_density_work *) calloc(1, sizeof(orc_density_work));
  const size_t n = (size_t) (nall > 0 ? nall : 1);
  w->nall = nall;
  w->prior = (double *) calloc(n * 2, sizeof(double));
  w->partial = (double *) calloc(n * 2, sizeof(double));
  w->G = (double *) calloc(n * 2, sizeof(double));
  w->S = (double *) calloc(n * 2, sizeof(double));
  w->cv = (double *) calloc(n * 2, sizeof(double));
  w->fpart = (double *) calloc(n * 3, sizeof(double)); /* mode 0 scatter target incl. ghosts */
  return w;
}

void orc_density_work_destroy(orc_density_work *w)
{
  if (!w) return;
  free(w->prior); free(w->partial); free(w->G); free(w->S); free(w->cv); free(w->fpart);
  free(w);
}

int orc_pair_density_check(orc_pair *p, const orc_atoms *a)
{
  const int nall = a->nlocal + a->nghost;
  if (p->style != ORC_STYLE_BETHE_DENSITY) {
    strcpy(p->errmsg, "orc_pair_density_compute needs style table_ucg_bethe_density");
    return 1;
  }
  for (int i = 0; i < nall; i++) {
    int t = a->type[i];
    if (t < 1 || t > p->n_actual || p->n_states_per_type[t] != 2) {
      strcpy(p->errmsg, "only 2-state types are live (Scenario 4)");
      return 1;
    }
  }
  return 0;
}

void orc_pair_density_pass1(orc_pair *p, orc_atoms *a, const orc_list *l, orc_density_work *w)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  const int nlocal = a->nlocal, nall = a->nlocal + a->nghost;
  const double kT = p->kT;
  const double *x = a->x;
  double *f = a->f;
  double *prior = w->prior, *partial = w->partial, *G = w->G, *S = w->S, *cv = w->cv, *fpart = w->fpart;
  (void) nt; (void) ms; (void) nlocal; (void) nall; (void) kT; (void) x; (void) f;
  (void) prior; (void) partial; (void) G; (void) S; (void) cv; (void) fpart;
  /* ---- pass 1 */
  for (int ii = 0; ii < l->inum; ii++) {
    const int i = l->ilist[ii];
    const int itype = a->type[i];
    if (p->use_density[itype] == 1) {
      double rho = 0.0;
      const int *row = l->neigh + l->first[ii];
      for (int jj = 0; jj < l->numneigh[ii]; jj++) {
        const int j = row[jj] & ORC_NEIGHMASK;
        const int jtype = a->type[j];
        const double delx = x[3 * i] - x[3 * j], dely = x[3 * i + 1] - x[3 * j + 1], delz = x[3 * i + 2] - x[3 * j + 2];
        const double rsq = delx * delx + dely * dely + delz * delz;
        if (rsq < p->cutsq[itype * nt + jtype]) rho += prox(p, itype, sqrt(rsq));
      }
      /* threshold_prob_and_partial_from_cv :107-113 */
      double th = orc_tanh((rho - p->cv_thresholds[itype]) / (0.1 * p->cv_thresholds[itype]));
      prior[2 * i] = 0.5 + 0.5 * th;
      partial[2 * i] = 0.5 * (1.0 - th * th) / (0.1 * p->cv_thresholds[itype]);
      prior[2 * i + 1] = 1.0 - prior[2 * i];
      partial[2 * i + 1] = -partial[2 * i];
    } else {
      double den = 0.0;
      for (int si = 0; si < 2; si++) {
        prior[2 * i + si] = orc_exp(-p->chem_pot[p->formal_from_actual[itype * ms + si]] / kT);
        den += prior[2 * i + si];
      }
      for (int si = 0; si < 2; si++) prior[2 * i + si] /= den;
    }
  }
}

void orc_pair_density_pass2(orc_pair *p, orc_atoms *a, const orc_list *l, int mode, int eflag, int vflag, orc_density_work *w,
                            orc_ev *ev)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  const int nlocal = a->nlocal, nall = a->nlocal + a->nghost;
  const double kT = p->kT;
  const double *x = a->x;
  double *f = a->f;
  double *prior = w->prior, *partial = w->partial, *G = w->G, *S = w->S, *cv = w->cv, *fpart = w->fpart;
  (void) nt; (void) ms; (void) nlocal; (void) nall; (void) kT; (void) x; (void) f;
  (void) prior; (void) partial; (void) G; (void) S; (void) cv; (void) fpart;
  /* ---- pass 2 */
  for (int ii = 0; ii < l->inum; ii++) {
    const int i = l->ilist[ii];
    const int itype = a->type[i];
    const int *row = l->neigh + l->first[ii];
    const int jnum = l->numneigh[ii];
    const double jnum_f = 1. - jnum;
    a->num_ucgstates[i] = p->n_states_per_type[itype];
    if (p->use_density[itype]) {
      for (int si = 0; si < 2; si++) {
        if (p->use_state_entropy[itype]) G[2 * i + si] -= kT * orc_log(prior[2 * i + si]) * jnum_f;
        G[2 * i + si] -= p->chem_pot[p->formal_from_actual[itype * ms + si]];
        S[2 * i + si] -= p->chem_pot[p->formal_from_actual[itype * ms + si]] / kT;
      }
    }
    double fx = 0.0, fy = 0.0, fz = 0.0;
    for (int jj = 0; jj < jnum; jj++) {
      int j = row[jj];
      const double factor_lj = p->special_lj[(j >> ORC_SBBITS) & 3];
      j &= ORC_NEIGHMASK;
      const int jtype = a->type[j];
      const int jstate = a->ucgstate[j];
      const double delx = x[3 * i] - x[3 * j], dely = x[3 * i + 1] - x[3 * j + 1], delz = x[3 * i + 2] - x[3 * j + 2];
      const double rsq = delx * delx + dely * dely + delz * delz;
      if (!(rsq < p->cutsq[itype * nt + jtype])) continue;
      dquad q;
      int rc = deval4(p, itype, jtype, rsq, factor_lj, &q);
      if (rc) {
        if (!ev->err) { ev->err = rc; ev->err_i = i; ev->err_j = j; }
        continue;
      }
      for (int si = 0; si < 2; si++) S[2 * i + si] -= q.u[si][jstate] / kT;
      double p00, p01, p10, p11;
      dclosure(p, &q, prior[2 * i + 1], prior[2 * j + 1], &p00, &p01, &p10, &p11);
      double evdwl = p00 * q.u[0][0] + p01 * q.u[0][1] + p10 * q.u[1][0] + p11 * q.u[1][1];
      double fpair = p00 * q.fp[0][0] + p01 * q.fp[0][1] + p10 * q.fp[1][0] + p11 * q.fp[1][1];
      if (j < nlocal) {
        evdwl = evdwl * 0.5;
        fpair = fpair * 0.5;
      }
      if (mode == 0) {
        fpart[3 * i] += delx * fpair; fpart[3 * i + 1] += dely * fpair; fpart[3 * i + 2] += delz * fpair;
        if (j < nlocal) { fpart[3 * j] -= delx * fpair; fpart[3 * j + 1] -= dely * fpair; fpart[3 * j + 2] -= delz * fpair; }
      } else {
        fx += delx * fpair; fy += dely * fpair; fz += delz * fpair;
        if (j < nlocal) {
          /* what j's own visit of this pair sends to i: the closure with the roles swapped */
          dquad qt;
          for (int sa = 0; sa < 2; sa++)
            for (int sb = 0; sb < 2; sb++) { qt.u[sa][sb] = q.u[sb][sa]; qt.fp[sa][sb] = q.fp[sb][sa]; }
          double t00, t01, t10, t11;
          dclosure(p, &qt, prior[2 * j + 1], prior[2 * i + 1], &t00, &t01, &t10, &t11);
          double fpj = t00 * qt.fp[0][0] + t01 * qt.fp[0][1] + t10 * qt.fp[1][0] + t11 * qt.fp[1][1];
          fpj = fpj * 0.5;
          const double djx = x[3 * j] - x[3 * i], djy = x[3 * j + 1] - x[3 * i + 1], djz = x[3 * j + 2] - x[3 * i + 2];
          fx -= djx * fpj; fy -= djy * fpj; fz -= djz * fpj;
        }
      }
      /* ev_tally with newton off: half per owned end */
      if (eflag) ev->eng_vdwl += (j < nlocal) ? evdwl : 0.5 * evdwl;
      if (vflag) {
        const double sc = (j < nlocal) ? 1.0 : 0.5;
        ev->virial[0] += sc * (delx * delx * fpair); ev->virial[1] += sc * (dely * dely * fpair);
        ev->virial[2] += sc * (delz * delz * fpair); ev->virial[3] += sc * (delx * dely * fpair);
        ev->virial[4] += sc * (delx * delz * fpair); ev->virial[5] += sc * (dely * delz * fpair);
      }
      if (p->use_density[itype] == 1) {
        G[2 * i] -= (q.u[1][0] - q.u[0][0] + kT * orc_log(p10 / p00));
        G[2 * i + 1] -= (q.u[1][1] - q.u[0][1] + kT * orc_log(p11 / p01));
      }
    }
    if (mode == 1) { f[3 * i] = fx; f[3 * i + 1] = fy; f[3 * i + 2] = fz; }
  }
  if (mode == 0)
    for (int i = 0; i < nlocal; i++) { f[3 * i] = fpart[3 * i]; f[3 * i + 1] = fpart[3 * i + 1]; f[3 * i + 2] = fpart[3 * i + 2]; }

  /* posterior (:678-696) and CV forces of the owned beads */
  for (int ii = 0; ii < l->inum; ii++) {
    const int i = l->ilist[ii];
    const int itype = a->type[i];
    double e0 = orc_exp(S[2 * i]), e1 = orc_exp(S[2 * i + 1]);
    double den = 0.0;
    den += e0;
    den += e1;
    a->ucgp[i] = e1 / den;
    a->scores[2 * i] = S[2 * i];
    a->scores[2 * i + 1] = S[2 * i + 1];
    if (p->use_density[itype] == 1) {
      cv[2 * i] = G[2 * i] * partial[2 * i];
      cv[2 * i + 1] = G[2 * i + 1] * partial[2 * i + 1];
    }
  }
}

void orc_pair_density_pass3(orc_pair *p, orc_atoms *a, const orc_list *l, int mode, int vflag, const int *ghost_src,
                            orc_density_work *w, orc_ev *ev)
{
  const int nt = p->n_formal + 1, ms = p->max_states;
  const int nlocal = a->nlocal, nall = a->nlocal + a->nghost;
  const double kT = p->kT;
  const double *x = a->x;
  double *f = a->f;
  double *prior = w->prior, *partial = w->partial, *G = w->G, *S = w->S, *cv = w->cv, *fpart = w->fpart;
  (void) nt; (void) ms; (void) nlocal; (void) nall; (void) kT; (void) x; (void) f;
  (void) prior; (void) partial; (void) G; (void) S; (void) cv; (void) fpart;
  /* ---- pass 3: back-force of the density CV (:698-733) */
  memset(fpart, 0, sizeof(double) * 3 * (size_t) nall);
  for (int ii = 0; ii < l->inum; ii++) {
    const int i = l->ilist[ii];
    const int itype = a->type[i];
    const int *row = l->neigh + l->first[ii];
    const int jnum = l->numneigh[ii];
    if (mode == 0) {
      /* the reference's loop shape: state outer, neighbours inner, scatter to j */
      if (p->use_density[itype] == 1) {
        for (int si = 0; si < 2; si++) {
          const double cv_force = cv[2 * i + si];
          for (int jj = 0; jj < jnum; jj++) {
            const int j = row[jj] & ORC_NEIGHMASK;
            const int jtype = a->type[j];
            const double delx = x[3 * i] - x[3 * j], dely = x[3 * i + 1] - x[3 * j + 1], delz = x[3 * i + 2] - x[3 * j + 2];
            const double rsq = delx * delx + dely * dely + delz * delz;
            if (rsq < p->cutsq[itype * nt + jtype]) {
              const double distance = sqrt(rsq);
              const double w = p->density_proximity_as_shipped ? prox(p, itype, distance) : prox_der(p, itype, distance);
              const double fpair = cv_force * w / distance;
              f[3 * i] += fpair * delx; f[3 * i + 1] += fpair * dely; f[3 * i + 2] += fpair * delz;
              fpart[3 * j] -= fpair * delx; fpart[3 * j + 1] -= fpair * dely; fpart[3 * j + 2] -= fpair * delz;
              if (vflag) {
                ev->virial[0] += delx * delx * fpair; ev->virial[1] += dely * dely * fpair; ev->virial[2] += delz * delz * fpair;
                ev->virial[3] += delx * dely * fpair; ev->virial[4] += delx * delz * fpair; ev->virial[5] += dely * delz * fpair;
              }
            }
          }
        }
      }
    } else {
      /* canonical: one sweep of the row; per neighbour first i's own CV force (state 0, 1), then
         what the neighbour's CV force (state 0, 1) sends back to i -- owned or ghost neighbour alike */
      double fx = f[3 * i], fy = f[3 * i + 1], fz = f[3 * i + 2];
      for (int jj = 0; jj < jnum; jj++) {
        const int j = row[jj] & ORC_NEIGHMASK;
        const int jtype = a->type[j];
        const double delx = x[3 * i] - x[3 * j], dely = x[3 * i + 1] - x[3 * j + 1], delz = x[3 * i + 2] - x[3 * j + 2];
        const double rsq = delx * delx + dely * dely + delz * delz;
        const double distance = sqrt(rsq);
        if (p->use_density[itype] == 1 && rsq < p->cutsq[itype * nt + jtype]) {
          const double w = p->density_proximity_as_shipped ? prox(p, itype, distance) : prox_der(p, itype, distance);
          for (int si = 0; si < 2; si++) {
            const double fpair = cv[2 * i + si] * w / distance;
            fx += fpair * delx; fy += fpair * dely; fz += fpair * delz;
            if (vflag) {
              ev->virial[0] += delx * delx * fpair; ev->virial[1] += dely * dely * fpair; ev->virial[2] += delz * delz * fpair;
              ev->virial[3] += delx * dely * fpair; ev->virial[4] += delx * delz * fpair; ev->virial[5] += dely * delz * fpair;
            }
          }
        }
        if (p->use_density[jtype] == 1 && rsq < p->cutsq[jtype * nt + itype]) {
          const double djx = x[3 * j] - x[3 * i], djy = x[3 * j + 1] - x[3 * i + 1], djz = x[3 * j + 2] - x[3 * i + 2];
          const double w = p->density_proximity_as_shipped ? prox(p, jtype, distance) : prox_der(p, jtype, distance);
          for (int sj = 0; sj < 2; sj++) {
            const double fpair = cv[2 * j + sj] * w / distance;
            fx -= fpair * djx; fy -= fpair * djy; fz -= fpair * djz;
          }
        }
      }
      f[3 * i] = fx; f[3 * i + 1] = fy; f[3 * i + 2] = fz;
    }
  }
  if (mode == 0) {
    /* scatter targets: owned j directly, ghost j through the reverse sum into its owner */
    for (int i = 0; i < nlocal; i++) { f[3 * i] += fpart[3 * i]; f[3 * i + 1] += fpart[3 * i + 1]; f[3 * i + 2] += fpart[3 * i + 2]; }
    for (int g = 0; g < a->nghost; g++) {
      const int src = ghost_src[g];
      for (int d = 0; d < 3; d++) f[3 * src + d] += fpart[3 * (nlocal + g) + d];
    }
  }
}

int orc_pair_density_compute(orc_pair *p, orc_atoms *a, const orc_list *l, int mode, int eflag, int vflag,
                             const int *ghost_src, orc_ev *ev)
{
  const int nlocal = a->nlocal, nall = a->nlocal + a->nghost;
  memset(ev, 0, sizeof(*ev));
  if (orc_pair_density_check(p, a)) return 1;
  orc_density_work *w = orc_density_work_create(nall);
  double *prior = w->prior, *partial = w->partial, *cv = w->cv;
  orc_pair_density_pass1(p, a, l, w);
  /* forward communication of the priors (fix of App. B #7) */
  for (int g = 0; g < a->nghost; g++) {
    const int src = ghost_src[g];
    prior[2 * (nlocal + g)] = prior[2 * src];
    prior[2 * (nlocal + g) + 1] = prior[2 * src + 1];
    partial[2 * (nlocal + g)] = partial[2 * src];
    partial[2 * (nlocal + g) + 1] = partial[2 * src + 1];
  }

  orc_pair_density_pass2(p, a, l, mode, eflag, vflag, w, ev);
  for (int g = 0; g < a->nghost; g++) {
    cv[2 * (nlocal + g)] = cv[2 * ghost_src[g]];
    cv[2 * (nlocal + g) + 1] = cv[2 * ghost_src[g] + 1];
  }

  orc_pair_density_pass3(p, a, l, mode, vflag, ghost_src, w, ev);
  orc_density_work_destroy(w);
  return ev->err ? 2 : 0;
}
