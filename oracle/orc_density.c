/* orc_density.c -- oracle compute() for table_ucg_bethe_density + small accessors.
 * TEST INFRASTRUCTURE (see orc.h). */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_pair_density_compute(orc_pair *p, orc_atoms *a, const orc_list *l, int eflag, int vflag,
                             orc_ev *ev);

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

int orc_pair_density_compute(orc_pair *p, orc_atoms *a, const orc_list *l, int eflag, int vflag,
                             orc_ev *ev)
{
  (void) a; (void) l; (void) eflag; (void) vflag;
  memset(ev, 0, sizeof(*ev));
  strcpy(p->errmsg, "table_ucg_bethe_density oracle not built yet");
  return 1;
}
