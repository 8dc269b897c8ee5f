/* orc_pair.c -- oracle pair-style setup (settings / coeff / init) shared by the
 * three UCG pair styles.  TEST INFRASTRUCTURE (see orc.h).
 *
 * Restates
 *   read_state_settings  UCG/pair_table_ucgld.cpp:565-652
 *                        UCG/pair_table_ucg_bethe_density.cpp:778-893 (density variant)
 *   settings             UCG/pair_table_ucgld.cpp:654-716
 *                        UCG/pair_table_ucg_bethe.cpp:746-886 (method / pseudo / prior keywords)
 *   coeff                UCG/pair_table_ucgld.cpp:719-865
 *   init_style/init_one  UCG/pair_table_ucgld.cpp:867-895, UCG/pair_table_ucg_bethe.cpp:1038-1088
 * plus upstream Pair::init()'s loop "for i<=j: cutsq[i][j]=cutsq[j][i]=init_one(i,j)^2".
 *
 * Deliberate deviations from the shipped text:
 *   #22 header line order is "n_actual n_formal max_states" (the code, not the comment)
 *   #25/#10 the density parser's out-of-bounds init loop and uninitialised ntables are not reproduced
 *   the settings file is tokenised on any whitespace (the reference uses strtok(" ") and
 *   then strcmp()s a token that may still carry the newline)
 */
#include "orc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXLINE 1024

static int fail(orc_pair *p, const char *msg)
{
  strncpy(p->errmsg, msg, sizeof(p->errmsg) - 1);
  p->errmsg[sizeof(p->errmsg) - 1] = '\0';
  return 1;
}

orc_pair *orc_pair_create(int style)
{
  orc_pair *p = (orc_pair *) calloc(1, sizeof(orc_pair));
  p->style = style;
  p->special_lj[0] = p->special_lj[1] = p->special_lj[2] = p->special_lj[3] = 1.0;
  /* UCG/pair_table_ucg_bethe.cpp:756-759 */
  p->pseudo_flag = 0;
  p->prior_flag = ORC_PRIOR_UCGL;
  p->method_flag = ORC_METHOD_BETHE;
  p->noise_level = 0.0;
  return p;
}

static void free_tables(orc_pair *p)
{
  for (int m = 0; m < p->ntables; m++) orc_table_free(&p->tables[m]);
  free(p->tables);
  p->tables = NULL;
  p->ntables = 0;
}

static void free_alloc(orc_pair *p)
{
  free(p->setflag); free(p->cutsq); free(p->tabindex);
  p->setflag = NULL; p->cutsq = NULL; p->tabindex = NULL;
  p->allocated = 0;
}

static void free_settings(orc_pair *p)
{
  free(p->n_states_per_type); free(p->formal_from_actual); free(p->actual_from_formal);
  free(p->chem_pot); free(p->prior_prob_from_type);
  free(p->use_density); free(p->use_state_entropy); free(p->cv_thresholds); free(p->threshold_radii);
  p->n_states_per_type = p->formal_from_actual = p->actual_from_formal = NULL;
  p->chem_pot = p->prior_prob_from_type = NULL;
  p->use_density = p->use_state_entropy = NULL;
  p->cv_thresholds = p->threshold_radii = NULL;
}

void orc_pair_destroy(orc_pair *p)
{
  if (!p) return;
  free_tables(p);
  free_alloc(p);
  free_settings(p);
  free(p);
}

const char *orc_pair_error(const orc_pair *p) { return p->errmsg; }

static int read_state_settings(orc_pair *p, const char *file)
{
  char line[MAXLINE];
  FILE *fp = fopen(file, "r");
  if (!fp) {
    char msg[400];
    snprintf(msg, sizeof msg, "Cannot open file %s", file);
    return fail(p, msg);
  }
  if (!fgets(line, MAXLINE, fp)) { fclose(fp); return fail(p, "Unexpected end of RLEUCG state settings file"); }
  if (sscanf(line, "%d %d %d", &p->n_actual, &p->n_formal, &p->max_states) != 3) {
    fclose(fp);
    return fail(p, "Bad header line in UCG state settings file");
  }
  if (p->max_states < 2) p->max_states = 2;
  free_settings(p);
  const int na = p->n_actual, nf = p->n_formal, ms = p->max_states;
  p->n_states_per_type = (int *) calloc((size_t) na + 1, sizeof(int));
  p->actual_from_formal = (int *) calloc((size_t) nf + 1, sizeof(int));
  p->chem_pot = (double *) calloc((size_t) nf + 1, sizeof(double));
  p->formal_from_actual = (int *) calloc((size_t) (na + 1) * ms, sizeof(int));
  p->prior_prob_from_type = (double *) calloc((size_t) (na + 1) * ms, sizeof(double));
  p->use_density = (int *) calloc((size_t) na + 1, sizeof(int));
  p->use_state_entropy = (int *) calloc((size_t) na + 1, sizeof(int));
  p->cv_thresholds = (double *) calloc((size_t) na + 1, sizeof(double));
  p->threshold_radii = (double *) calloc((size_t) na + 1, sizeof(double));

  for (int i = 1; i <= na; i++) {
    if (!fgets(line, MAXLINE, fp)) { fclose(fp); return fail(p, "Unexpected end of UCG state settings file"); }
    int this_type = 0;
    sscanf(line, "%d %d", &this_type, &p->n_states_per_type[i]);
    if (p->n_states_per_type[i] < 1 || p->n_states_per_type[i] > 2) {
      fclose(fp);
      return fail(p, "Invalid number of states for atom type. Only 1 or 2 states are allowed.");
    } else if (this_type != i) {
      fclose(fp);
      return fail(p, "Please write orderly. Invalid atom type in UCG state settings file.");
    }
    if (p->n_states_per_type[i] == 2) {
      if (!fgets(line, MAXLINE, fp)) { fclose(fp); return fail(p, "Unexpected end of UCG state settings file"); }
      char *save = NULL;
      char *tok = strtok_r(line, " \t\r\n", &save);
      for (int j = 0; j < 2; j++) {
        if (!tok) { fclose(fp); return fail(p, "Not enough formal types specified for atom type."); }
        int ft = atoi(tok);
        if (ft < 0 || ft > nf) { fclose(fp); return fail(p, "Formal type out of range in UCG state settings file"); }
        p->formal_from_actual[i * ms + j] = ft;
        p->actual_from_formal[ft] = i;
        tok = strtok_r(NULL, " \t\r\n", &save);
      }
      if (p->style == ORC_STYLE_BETHE_DENSITY) {
        /* ...density.cpp:854-871 */
        if (!tok) { fclose(fp); return fail(p, "Missing state type for atom type."); }
        char state_type[64];
        strncpy(state_type, tok, 63); state_type[63] = '\0';
        tok = strtok_r(NULL, " \t\r\n", &save);
        if (!tok) { fclose(fp); return fail(p, "Missing entropy specification for atom type."); }
        if (strcmp(tok, "entropy") == 0) p->use_state_entropy[i] = 1;
        else if (strcmp(tok, "no_entropy") == 0) p->use_state_entropy[i] = 0;
        else { fclose(fp); return fail(p, "Unknown entropy specification. Use 'entropy' or 'no_entropy'."); }
        if (strcmp(state_type, "density") == 0) {
          p->use_density[i] = 1;
          if (!fgets(line, MAXLINE, fp)) { fclose(fp); return fail(p, "Unexpected end of RLEUCG state settings file"); }
          sscanf(line, "%lg %lg", &p->cv_thresholds[i], &p->threshold_radii[i]);
        }
      }
      if (!fgets(line, MAXLINE, fp)) { fclose(fp); return fail(p, "Unexpected end of UCG state settings file"); }
      save = NULL;
      tok = strtok_r(line, " \t\r\n", &save);
      for (int j = 0; j < 2; j++) {
        if (!tok) { fclose(fp); return fail(p, "Not enough chemical potentials specified for atom type."); }
        p->chem_pot[p->formal_from_actual[i * ms + j]] = strtod(tok, NULL);
        tok = strtok_r(NULL, " \t\r\n", &save);
      }
    }
  }
  fclose(fp);
  return 0;
}

int orc_pair_settings(orc_pair *p, int narg, const char **arg)
{
  if (narg < 3) return fail(p, "Illegal pair_style command: expected <style> <N> <state settings file>");
  if (strcmp(arg[0], "lookup") == 0) p->tabstyle = ORC_LOOKUP;
  else if (strcmp(arg[0], "linear") == 0) p->tabstyle = ORC_LINEAR;
  else if (strcmp(arg[0], "spline") == 0) p->tabstyle = ORC_SPLINE;
  else if (strcmp(arg[0], "bitmap") == 0) p->tabstyle = ORC_BITMAP;
  else return fail(p, "Unknown table style in pair_style command");
  p->tablength = atoi(arg[1]);
  if (p->tablength < 2) return fail(p, "Illegal number of pair table entries");

  p->pseudo_flag = 0;
  p->prior_flag = ORC_PRIOR_UCGL;
  p->method_flag = ORC_METHOD_BETHE;
  p->noise_level = 0.0;

  if (read_state_settings(p, arg[2])) return 1;

  int iarg = 3;
  while (iarg < narg) {
    if (!strcmp(arg[iarg], "ewald") || !strcmp(arg[iarg], "pppm") || !strcmp(arg[iarg], "msm") ||
        !strcmp(arg[iarg], "dispersion") || !strcmp(arg[iarg], "tip4p")) {
      /* KSpace compatibility flags: accepted, no effect on this path */
    } else if (p->style == ORC_STYLE_BETHE && strcmp(arg[iarg], "method") == 0) {
      iarg++;
      if (iarg >= narg) return fail(p, "Missing argument for pair_style table_ucg_bethe method");
      if (!strcmp(arg[iarg], "mf") || !strcmp(arg[iarg], "meanfield")) p->method_flag = ORC_METHOD_MF;
      else if (!strcmp(arg[iarg], "bethe") || !strcmp(arg[iarg], "Bethe")) p->method_flag = ORC_METHOD_BETHE;
      else return fail(p, "Unknown argument for pair_style table_ucg_bethe method, please write mf or bethe");
    } else if (p->style == ORC_STYLE_BETHE && strcmp(arg[iarg], "pseudo") == 0) {
      iarg++;
      if (iarg >= narg) return fail(p, "Missing argument for pair_style table_ucg_bethe pseudo");
      if (!strcmp(arg[iarg], "yes")) p->pseudo_flag = 0;
      else if (!strcmp(arg[iarg], "no")) p->pseudo_flag = 1;
      else return fail(p, "Unknown argument for pair_style table_ucg_bethe pseudo, please write yes or no");
    } else if (p->style == ORC_STYLE_BETHE && strcmp(arg[iarg], "prior") == 0) {
      iarg++;
      if (iarg >= narg) return fail(p, "Missing argument for pair_style table_ucg_bethe");
      if (!strcmp(arg[iarg], "chemical_potential")) {
        iarg += 1;
        if (iarg >= narg) {
          p->prior_flag = ORC_PRIOR_CHEMPOT;
          iarg -= 1;
        } else if (!strcmp(arg[iarg], "noise")) {
          p->prior_flag = ORC_PRIOR_CHEMPOT_NOISE;
          iarg += 1;
          if (iarg >= narg) return fail(p, "Missing argument for prior chemical_potential noise: noise level must be set");
          p->noise_level = strtod(arg[iarg], NULL);
          if (p->noise_level <= 0.0) p->noise_level = 0.0;
          iarg += 1;
          if (iarg >= narg) return fail(p, "Missing argument for prior chemical_potential noise: random seed must be set");
          p->seed = atoi(arg[iarg]);
          if (p->seed <= 0) p->seed = -p->seed + 1;
          orc_ranmars_init(&p->random, p->seed + 0);
          p->have_random = 1;
        }
        /* NOTE (as shipped, :838-858): "prior chemical_potential <other keyword>" leaves
           prior_flag at its previous value and swallows <other keyword>. */
      } else if (!strcmp(arg[iarg], "ucgl")) {
        p->prior_flag = ORC_PRIOR_UCGL;
      } else {
        return fail(p, "Unknown argument for pair_style table_ucg_bethe prior, please write chemical_potential or ucgl");
      }
    } else if (p->style == ORC_STYLE_BETHE) {
      /* as shipped (:796-868) unknown keywords are silently ignored by table_ucg_bethe */
    } else {
      return fail(p, "Unknown pair_style table keyword");
    }
    iarg++;
  }

  free_tables(p);
  free_alloc(p);
  return 0;
}

static void allocate(orc_pair *p)
{
  const int nt = p->n_formal + 1;
  p->allocated = 1;
  p->setflag = (int *) calloc((size_t) nt * nt, sizeof(int));
  p->cutsq = (double *) calloc((size_t) nt * nt, sizeof(double));
  p->tabindex = (int *) calloc((size_t) nt * nt, sizeof(int));
}

/* utils::bounds for the forms the UCG decks use: "N", "*", "N*", "*N", "M*N" */
static int parse_bounds(const char *s, int nmin, int nmax, int *lo, int *hi)
{
  const char *star = strchr(s, '*');
  if (!star) {
    *lo = *hi = atoi(s);
  } else if (strlen(s) == 1) {
    *lo = nmin; *hi = nmax;
  } else if (star == s) {
    *lo = nmin; *hi = atoi(s + 1);
  } else if (*(star + 1) == '\0') {
    *lo = atoi(s); *hi = nmax;
  } else {
    *lo = atoi(s); *hi = atoi(star + 1);
  }
  if (*lo < nmin || *hi > nmax || *lo > *hi) return 1;
  return 0;
}

int orc_pair_coeff(orc_pair *p, int ntypes, int narg, const char **arg)
{
  if (narg < 7) {
    if (narg == 6) return fail(p, "This pair style requires explicit definition of cutoff for each table.");
    return fail(p, "Too few arguments.");
  }
  if (!p->n_states_per_type) return fail(p, "pair_coeff before pair_style");
  if (!p->allocated) allocate(p);
  const int nt = p->n_formal + 1, ms = p->max_states;

  int ilo, ihi, jlo, jhi;
  if (parse_bounds(arg[0], 1, ntypes, &ilo, &ihi)) return fail(p, "Invalid type range in pair_coeff");
  if (parse_bounds(arg[1], 1, ntypes, &jlo, &jhi)) return fail(p, "Invalid type range in pair_coeff");

  const int Ns_i = atoi(arg[2]);
  const int Ns_j = atoi(arg[3]);
  /* "Just serves as a check" loops run over [lo,hi) as shipped (:766-775) */
  for (int t = ilo; t < ihi; t++)
    if (t <= p->n_actual && Ns_i != p->n_states_per_type[t])
      return fail(p, "Number of states for atom type does not match the number of states in the settings file.");
  for (int t = jlo; t < jhi; t++)
    if (t <= p->n_actual && Ns_j != p->n_states_per_type[t])
      return fail(p, "Number of states for atom type does not match the number of states in the settings file.");

  const int ntables_this = Ns_i * Ns_j;
  if (narg != 4 + 3 * ntables_this)
    return fail(p, "Incorrect number of arguments for pair_coeff command. Expected 4 + 3 * n_states_i * n_states_j arguments.");
  if (ihi > p->n_actual || jhi > p->n_actual)
    return fail(p, "pair_coeff I J must name ACTUAL types (<= n_actual of the state settings file)");

  int this_i = 4;
  for (int s_i = 0; s_i < Ns_i; s_i++) {
    for (int s_j = 0; s_j < Ns_j; s_j++) {
      p->tables = (orc_table *) realloc(p->tables, sizeof(orc_table) * (size_t) (p->ntables + 1));
      orc_table *tb = &p->tables[p->ntables];
      char err[400];
      if (orc_table_read(tb, arg[this_i], arg[this_i + 1], err, sizeof err)) return fail(p, err);
      double cut = strtod(arg[this_i + 2], NULL);
      if (orc_table_build(tb, p->tabstyle, p->tablength, cut, err, sizeof err)) {
        orc_table_free(tb);
        return fail(p, err);
      }
      int count = 0;
      for (int i = ilo; i <= ihi; i++) {
        for (int j = (jlo > i ? jlo : i); j <= jhi; j++) {
          int fi = p->formal_from_actual[i * ms + s_i];
          int fj = p->formal_from_actual[j * ms + s_j];
          if (fi == 0 || fj == 0) {
            p->ntables++;
            return fail(p, "Formal type not defined in pair_style command for actual type / state");
          }
          p->tabindex[fi * nt + fj] = p->ntables;
          p->setflag[fi * nt + fj] = 1;
          count++;
        }
      }
      p->ntables++;
      if (count == 0) return fail(p, "Illegal pair_coeff command");
      this_i += 3;
    }
  }
  return 0;
}

int orc_pair_init(orc_pair *p, int ntypes, double T, double boltz)
{
  if (!p->allocated) return fail(p, "All pair coeffs are not set");
  const int nt = p->n_formal + 1, ms = p->max_states;
  if (ntypes > p->n_formal) return fail(p, "atom->ntypes exceeds n_formal of the state settings file");
  p->T = T;
  p->kT = boltz * T;
  /* Pair::init(): for i<=j: init_one(i,j) */
  for (int i = 1; i <= ntypes; i++) {
    for (int j = i; j <= ntypes; j++) {
      if (p->setflag[i * nt + j] == 0) return fail(p, "All pair coeffs are not set");
      p->tabindex[j * nt + i] = p->tabindex[i * nt + j];
      double cut = p->tables[p->tabindex[i * nt + j]].cut;
      p->cutsq[i * nt + j] = p->cutsq[j * nt + i] = cut * cut;
    }
  }
  if (p->style == ORC_STYLE_BETHE) {
    /* UCG/pair_table_ucg_bethe.cpp:1056-1076 */
    double denomi = 0.0;
    for (int i = 1; i <= p->n_actual; i++) {
      if (p->n_states_per_type[i] == 0) continue;
      else if (p->n_states_per_type[i] == 1) {
        p->prior_prob_from_type[i * ms + 0] = 1.0;
      } else {
        for (int j = 0; j < p->n_states_per_type[i]; j++) {
          p->prior_prob_from_type[i * ms + j] = orc_exp(-p->chem_pot[p->formal_from_actual[i * ms + j]] / p->kT);
          denomi += p->prior_prob_from_type[i * ms + j];
        }
        for (int j = 0; j < p->n_states_per_type[i]; j++) p->prior_prob_from_type[i * ms + j] /= denomi;
        denomi = 0.0;
      }
    }
  }
  orc_pair_sum_scales(p);
  return 0;
}
