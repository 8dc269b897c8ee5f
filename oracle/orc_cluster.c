/* orc_cluster.c -- CPU restatement of fix cluster_switch (UCG/fix_cluster_switch.cpp).
 *
 * TEST INFRASTRUCTURE (see orc.h): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.  PARITY UNPINNED: the reference ships no fixture for this fix.
 *
 * Single rank (every MPI_Allreduce of the reference is the identity).  Where the reference's result
 * depends on the LOCAL INDEX ORDER of atoms -- the first switchable atom of a molecule decides its
 * initial state (:140-156), and confirm_molecule fills at most nSwitchPerMol slots per molecule in
 * index order (:804-857) -- this restatement walks the atoms in ascending TAG order, which is the
 * local order of a freshly read data file and does not depend on how the driver sorts its beads.
 * The debug logs cluster_assignment.log / state_assignment.log (:693-714) are not written; the same
 * data are available through orc_cs_arrays().
 */
#include "orc_cluster.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXLINE 1024

static int fail(orc_cluster_switch *cs, const char *msg)
{
  snprintf(cs->errmsg, sizeof cs->errmsg, "%s", msg);
  return 1;
}

/* words of a line after stripping a '#' comment; returns the number of words */
static int split_words(char *line, char **words, int maxwords)
{
  char *p = strchr(line, '#');
  if (p) *p = '\0';
  int n = 0;
  for (char *w = strtok(line, " \t\n\r\f"); w && n < maxwords; w = strtok(NULL, " \t\n\r\f")) words[n++] = w;
  return n;
}

/* read_file (:207-277).  The line number counts EVERY physical line, blank and comment lines too. */
static int read_rates(orc_cluster_switch *cs, const char *file, int ntypes)
{
  FILE *fp = fopen(file, "r");
  if (!fp) return fail(cs, "Cannot open file (rates)");
  char line[MAXLINE], *words[100];
  int lineNum = 0;
  while (fgets(line, MAXLINE, fp)) {
    lineNum++;
    const int nw = split_words(line, words, 100);
    if (nw == 0) continue;
    if (lineNum == 1) {
      cs->probON = atof(words[0]);
      if (cs->probON > 1.0) {
        fclose(fp);
        return fail(cs, "Incorrect probability in rates.txt files (fix cluster_switch)");
      }
      cs->probOFF = 1.0 - cs->probON;
    } else if (lineNum == 2) {
      cs->nSwitchTypes = atoi(words[0]);
      if (cs->nSwitchTypes > ntypes || cs->nSwitchTypes < 1) {
        fclose(fp);
        return fail(cs, "Incorrect number of atom switching types (fix cluster_switch)");
      }
      cs->typesON = (int *) calloc((size_t) cs->nSwitchTypes, sizeof(int));
      cs->typesOFF = (int *) calloc((size_t) cs->nSwitchTypes, sizeof(int));
    } else if (lineNum == 3 && cs->typesON) {
      for (int i = 0; i < cs->nSwitchTypes && i < nw; i++) cs->typesON[i] = atoi(words[i]);
    } else if (lineNum == 4 && cs->typesOFF) {
      for (int i = 0; i < cs->nSwitchTypes && i < nw; i++) cs->typesOFF[i] = atoi(words[i]);
    }
  }
  fclose(fp);
  if (!cs->typesON) return fail(cs, "rates file has no switching types (fix cluster_switch)");
  return 0;
}

/* read_contacts (:281-344): "<label> nContactTypes" / "<label> nAtomsPerContact" / pairs */
static int read_contacts(orc_cluster_switch *cs, const char *file)
{
  FILE *fp = fopen(file, "r");
  if (!fp) return fail(cs, "Cannot open file (contacts)");
  char line[MAXLINE], *words[100];
  int lineNum = 0;
  while (fgets(line, MAXLINE, fp)) {
    lineNum++;
    const int nw = split_words(line, words, 100);
    if (nw == 0) continue;
    if (lineNum == 1) {
      if (nw < 2) { fclose(fp); return fail(cs, "contacts file: line 1 needs a label and the number of contact types"); }
      cs->nContactTypes = atoi(words[1]);
    } else if (lineNum == 2) {
      if (nw < 2) { fclose(fp); return fail(cs, "contacts file: line 2 needs a label and the atoms per contact"); }
      cs->nAtomsPerContact = atoi(words[1]);
      if (cs->nContactTypes < 1 || cs->nAtomsPerContact < 1) { fclose(fp); return fail(cs, "contacts file: empty contact map"); }
      cs->contactMap = (int *) calloc((size_t) cs->nContactTypes * cs->nAtomsPerContact * 2, sizeof(int));
    } else if (cs->contactMap) {
      const int off = lineNum - 3;
      const int i = off / cs->nAtomsPerContact, j = off - i * cs->nAtomsPerContact;
      if (i >= cs->nContactTypes || nw < 2) { fclose(fp); return fail(cs, "contacts file: more pairs than declared"); }
      cs->contactMap[(i * cs->nAtomsPerContact + j) * 2 + 0] = atoi(words[0]);
      cs->contactMap[(i * cs->nAtomsPerContact + j) * 2 + 1] = atoi(words[1]);
    }
  }
  fclose(fp);
  if (!cs->contactMap) return fail(cs, "contacts file has no contact map");
  return 0;
}

static int cmp_tag(const void *a, const void *b)
{
  const int *x = (const int *) a, *y = (const int *) b;
  return (x[0] > y[0]) - (x[0] < y[0]);
}

/* owned atoms in ascending tag order: out[2*k] = tag, out[2*k+1] = local index */
static int *by_tag(const orc_atoms *a)
{
  int *o = (int *) malloc(sizeof(int) * 2 * (size_t) (a->nlocal + 1));
  for (int i = 0; i < a->nlocal; i++) {
    o[2 * i] = a->tag[i];
    o[2 * i + 1] = i;
  }
  qsort(o, (size_t) a->nlocal, 2 * sizeof(int), cmp_tag);
  return o;
}

static int check_arrays(orc_cluster_switch *cs)
{
  for (int i = 0; i <= cs->maxmol; i++)
    if (cs->mol_restrict[i] == 1 && !(cs->mol_state[i] == 1 || cs->mol_state[i] == 0))
      return fail(cs, "Communication of mol_state inconsistent: fix cluster_switch");
  return 0;
}

orc_cluster_switch *orc_cs_create(const orc_atoms *a, const int *molecule, int ntypes, int groupbit, int mol_seed,
                                  int mol_offset, double cutoff, int seed, int switchFreq, const char *rateFile,
                                  const char *contactFile, long long ntimestep)
{
  /* constructor (:37-181) */
  orc_cluster_switch *cs = (orc_cluster_switch *) calloc(1, sizeof(orc_cluster_switch));
  cs->mol_seed = mol_seed;
  cs->mol_offset = mol_offset;
  cs->cutsq = cutoff * cutoff;
  cs->switchFreq = switchFreq;
  cs->groupbit = groupbit;
  orc_ranpark_init(&cs->random_equal, seed);
  orc_ranpark_init(&cs->random_unequal, seed);
  cs->next_reneighbor = ntimestep + 1;
  if (read_rates(cs, rateFile, ntypes) || read_contacts(cs, contactFile)) return cs;

  int nmolatoms = 0, maxmol = -1, nSwitchPerMol = 0;
  for (int i = 0; i < a->nlocal; i++) {
    if (!(a->mask[i] & groupbit)) continue;
    if (molecule[i] > maxmol) maxmol = molecule[i];
    for (int j = 0; j < cs->nSwitchTypes; j++)
      if (a->type[i] == cs->typesON[j] || a->type[i] == cs->typesOFF[j]) {
        nmolatoms++;
        if (molecule[i] == mol_seed) nSwitchPerMol++;
      }
  }
  if (maxmol < 0) { fail(cs, "Selected group does not have any mols (fix cluster_switch)"); return cs; }
  if (nSwitchPerMol < 1) { fail(cs, "fix cluster_switch: molecule mol_seed has no switchable atoms (division by zero in the reference)"); return cs; }
  if (mol_seed < 0 || mol_seed > maxmol || mol_seed - mol_offset < 0 || mol_seed - mol_offset > maxmol) {
    fail(cs, "fix cluster_switch: mol_seed / mol_seed - mol_offset outside 0..maxmol (out-of-bounds write in the reference)");
    return cs;
  }
  cs->maxmol = maxmol;
  cs->nSwitchPerMol = nSwitchPerMol;
  cs->nmol = nmolatoms / nSwitchPerMol;
  const size_t nm = (size_t) maxmol + 1;
  cs->mol_restrict = (int *) malloc(nm * sizeof(int));
  cs->mol_state = (int *) malloc(nm * sizeof(int));
  cs->mol_accept = (int *) malloc(nm * sizeof(int));
  cs->mol_cluster = (int *) malloc(nm * sizeof(int));
  cs->mol_atoms = (int *) malloc(nm * (size_t) nSwitchPerMol * sizeof(int));
  for (size_t i = 0; i < nm; i++) cs->mol_restrict[i] = cs->mol_state[i] = cs->mol_accept[i] = cs->mol_cluster[i] = -1;
  for (size_t i = 0; i < nm * (size_t) nSwitchPerMol; i++) cs->mol_atoms[i] = -1;

  /* :140-156, in ascending tag order */
  int *ord = by_tag(a);
  for (int k = 0; k < a->nlocal; k++) {
    const int i = ord[2 * k + 1];
    if (!(a->mask[i] & groupbit)) continue;
    const int molID = molecule[i];
    for (int j = 0; j < cs->nSwitchTypes; j++) {
      if (a->type[i] == cs->typesON[j] && cs->mol_state[molID] == -1) {
        cs->mol_state[molID] = 1;
        if (molID != mol_seed && molID != (mol_seed - mol_offset)) cs->mol_restrict[molID] = 1;
      } else if (a->type[i] == cs->typesOFF[j] && cs->mol_state[molID] == -1) {
        cs->mol_state[molID] = 0;
        if (molID != mol_seed && molID != (mol_seed - mol_offset)) cs->mol_restrict[molID] = 1;
      }
    }
  }
  free(ord);
  check_arrays(cs);
  return cs;
}

void orc_cs_destroy(orc_cluster_switch *cs)
{
  if (!cs) return;
  free(cs->typesON); free(cs->typesOFF); free(cs->contactMap);
  free(cs->mol_restrict); free(cs->mol_state); free(cs->mol_accept); free(cs->mol_cluster); free(cs->mol_atoms);
  free(cs);
}

const char *orc_cs_error(const orc_cluster_switch *cs) { return cs->errmsg[0] ? cs->errmsg : NULL; }

static int switchable(const orc_cluster_switch *cs, int m) { return cs->mol_state[m] == 0 || cs->mol_state[m] == 1; }

/* the "offset partner" of a molecule (:629-646); -1 where the reference would index outside 0..maxmol */
static int partner(const orc_cluster_switch *cs, int m)
{
  const int p = switchable(cs, m) ? m - cs->mol_offset : m + cs->mol_offset;
  return (p < 0 || p > cs->maxmol) ? -1 : p;
}

#define IMIN(a, b) ((a) < (b) ? (a) : (b))

/* check_cluster (:551-719) in three phases, so that a decomposed run (orc_world) can reduce the labels over the ranks
   between sweeps the way the reference does (MPI_Allreduce MAX of the starting labels :587, MIN between sweeps :683):
     orc_cs_presence     the molecules with an in-group atom among these owned atoms (what :573-580 label)
     orc_cs_labels_init  the starting labels from the molecules present ANYWHERE (:573-599)
     orc_cs_sweep_local  sequential sweeps over these rows until one changes nothing; returns "changed at all"
     orc_cs_finalize     restrict / state flags of the seed's cluster (:675-690) */
void orc_cs_presence(const orc_cluster_switch *cs, const orc_atoms *a, const int *molecule, int *present)
{
  for (int i = 0; i < a->nlocal; i++)
    if ((a->mask[i] & cs->groupbit) && molecule[i] >= 0 && molecule[i] <= cs->maxmol) present[molecule[i]] = 1;
}

void orc_cs_labels_init(orc_cluster_switch *cs, const int *present, int *lab)
{
  const int maxmol = cs->maxmol;
  for (int i = 0; i <= maxmol; i++) lab[i] = -1;
  lab[cs->mol_seed] = cs->mol_seed;
  lab[cs->mol_seed - cs->mol_offset] = cs->mol_seed;
  for (int m = 0; m <= maxmol; m++)
    if (present[m]) lab[m] = m;
  for (int m = 0; m <= maxmol; m++)
    if (present[m] && switchable(cs, m)) {
      const int p = m - cs->mol_offset;
      if (p >= 0 && p <= maxmol) lab[p] = m;
    }
  cs->sweeps = 0;
}

int orc_cs_sweep_local(orc_cluster_switch *cs, const orc_atoms *a, const int *molecule, const orc_list *list, int *lab)
{
  const int nct = cs->nContactTypes, napc = cs->nAtomsPerContact;
  int any = 0;
  for (;;) {
    int done = 1;
    cs->sweeps++;
    for (int ii = 0; ii < list->inum; ii++) {
      const int i = list->ilist[ii];
      if (!(a->mask[i] & cs->groupbit)) continue;
      const int im = molecule[i], itype = a->type[i];
      const double xt = a->x[3 * i], yt = a->x[3 * i + 1], zt = a->x[3 * i + 2];
      const int *jl = list->neigh + list->first[ii];
      for (int jj = 0; jj < list->numneigh[ii]; jj++) {
        const int j = jl[jj] & ORC_NEIGHMASK;
        if (!(a->mask[j] & cs->groupbit)) continue;
        const int jm = molecule[j], jtype = a->type[j];
        if (lab[im] == lab[jm]) continue;
        int contact = 0;
        for (int m = 0; m < nct * napc && !contact; m++)
          if (cs->contactMap[2 * m] == itype && cs->contactMap[2 * m + 1] == jtype) contact = 1;
        if (!contact) continue;
        const double dx = xt - a->x[3 * j], dy = yt - a->x[3 * j + 1], dz = zt - a->x[3 * j + 2];
        const double rsq = dx * dx + dy * dy + dz * dz;
        if (rsq < cs->cutsq) {
          const int pi = partner(cs, im), pj = partner(cs, jm);
          int id = IMIN(lab[im], lab[jm]);
          if (pi >= 0) id = IMIN(lab[pi], id);
          if (pj >= 0) id = IMIN(lab[pj], id);
          lab[im] = lab[jm] = id;
          if (pi >= 0) lab[pi] = id;
          if (pj >= 0) lab[pj] = id;
          done = 0;
        }
      }
    }
    if (done) break;
    any = 1;
  }
  return any;
}

void orc_cs_finalize(orc_cluster_switch *cs, const int *lab)
{
  const int maxmol = cs->maxmol;
  memcpy(cs->mol_cluster, lab, sizeof(int) * ((size_t) maxmol + 1));
  /* :675-690 */
  const int clusterID = cs->mol_cluster[cs->mol_seed];
  cs->nCluster = 0.0;
  for (int i = 0; i <= maxmol; i++) {
    if (cs->mol_cluster[i] != -1) {
      if (switchable(cs, i)) {
        if (cs->mol_cluster[i] == clusterID) {
          cs->mol_restrict[i] = -1;
          cs->mol_state[i] = 1;
        } else
          cs->mol_restrict[i] = 1;
      }
      if (cs->mol_cluster[i] == clusterID) cs->nCluster += 1.0;
    }
  }
}

/* single rank: every reduction is the identity */
int orc_cs_check_cluster(orc_cluster_switch *cs, const orc_atoms *a, const int *molecule, const orc_list *list)
{
  const int maxmol = cs->maxmol;
  int *lab = (int *) malloc(sizeof(int) * ((size_t) maxmol + 1));
  int *present = (int *) calloc((size_t) maxmol + 1, sizeof(int));
  for (int ii = 0; ii < list->inum; ii++) {
    const int i = list->ilist[ii];
    if ((a->mask[i] & cs->groupbit) && molecule[i] >= 0 && molecule[i] <= maxmol) present[molecule[i]] = 1;
  }
  orc_cs_labels_init(cs, present, lab);
  orc_cs_sweep_local(cs, a, molecule, list, lab);
  orc_cs_finalize(cs, lab);
  free(present);
  free(lab);
  return 0;
}

/* confirm_molecule (:804-857) over the owned atoms in ascending tag order */
static int confirm_molecule(orc_cluster_switch *cs, const orc_atoms *a, const int *molecule, const int *ord, int molID)
{
  double sumState = 0.0;
  const double decisionBuffer = (double) cs->nSwitchPerMol / 2.0 - 1.0 + 0.01;
  int *slots = cs->mol_atoms + (size_t) molID * cs->nSwitchPerMol;
  for (int q = 0; q < a->nlocal; q++) {
    const int i = ord[2 * q + 1];
    if (molecule[i] != molID) continue;
    const int itype = a->type[i];
    for (int k = 0; k < cs->nSwitchTypes; k++) {
      if (itype == cs->typesON[k]) {
        for (int j = 0; j < cs->nSwitchPerMol; j++)
          if (slots[j] == -1) { slots[j] = i; sumState += 1.0; break; }
      } else if (itype == cs->typesOFF[k]) {
        for (int j = 0; j < cs->nSwitchPerMol; j++)
          if (slots[j] == -1) { slots[j] = i; sumState -= 1.0; break; }
      }
    }
  }
  if (sumState < (decisionBuffer * -1)) return -1;
  else if (sumState > decisionBuffer) return 1;
  return 0;
}

/* attempt_switch (:721-802), switch_flag (:860-885), gather_statistics (:899-935), in two phases: attempt_local decides the
   molecules THESE owned atoms make this rank the decision maker of (confirm_molecule's majority rule over its atoms, one
   RanPark draw each, ascending molecule id) and leaves the decisions in mol_accept; a decomposed run takes the maximum over
   the ranks (:793) before attempt_apply flips the types of the owned atoms and the molecules' states */
void orc_cs_attempt_local(orc_cluster_switch *cs, const orc_atoms *a, const int *molecule)
{
  const int maxmol = cs->maxmol;
  char *present = (char *) calloc((size_t) maxmol + 1, 1);
  for (int i = 0; i < a->nlocal; i++)
    if ((a->mask[i] & cs->groupbit) && molecule[i] >= 0 && molecule[i] <= maxmol) present[molecule[i]] = 1;
  for (int i = 0; i <= maxmol; i++) cs->mol_accept[i] = -1;
  for (size_t i = 0; i < ((size_t) maxmol + 1) * (size_t) cs->nSwitchPerMol; i++) cs->mol_atoms[i] = -1;
  int *ord = by_tag(a);
  for (int mID = 0; mID <= maxmol; mID++) {  /* std::map iteration = ascending molecule id */
    if (!present[mID]) continue;
    int confirmflag = 0;
    if (cs->mol_restrict[mID] == 1) confirmflag = confirm_molecule(cs, a, molecule, ord, mID);
    if (cs->mol_accept[mID] == -1 && confirmflag != 0) {
      const double checkProb = (cs->mol_state[mID] == 0) ? cs->probON : cs->probOFF;
      const double r = orc_ranpark_uniform(&cs->random_unequal);
      cs->mol_accept[mID] = (r < checkProb) ? 1 : 0;
    }
  }
  free(ord);
  free(present);
}

int orc_cs_attempt_apply(orc_cluster_switch *cs, orc_atoms *a)
{
  const int maxmol = cs->maxmol;
  /* gather_statistics, before the states flip */
  for (int i = 0; i <= maxmol; i++) {
    if (cs->mol_restrict[i] != 1) continue;
    cs->stats[0] += 1.0;
    if (cs->mol_state[i] == 0) {
      cs->stats[2] += 1.0;
      if (cs->mol_accept[i] == 1) { cs->stats[1] += 1.0; cs->stats[4] += 1.0; }
    } else if (cs->mol_state[i] == 1) {
      cs->stats[3] += 1.0;
      if (cs->mol_accept[i] == 1) { cs->stats[1] += 1.0; cs->stats[5] += 1.0; }
    }
  }
  if (check_arrays(cs)) return 1;

  for (int i = 0; i <= maxmol; i++) {
    if (cs->mol_accept[i] != 1) continue;
    for (int j = 0; j < cs->nSwitchPerMol; j++) {
      const int t = cs->mol_atoms[(size_t) i * cs->nSwitchPerMol + j];
      if (cs->mol_state[i] == 0 && t > -1) {
        for (int k = 0; k < cs->nSwitchTypes; k++)
          if (a->type[t] == cs->typesOFF[k]) a->type[t] = cs->typesON[k];
      } else if (cs->mol_state[i] == 1 && t > -1) {
        for (int k = 0; k < cs->nSwitchTypes; k++)
          if (a->type[t] == cs->typesON[k]) a->type[t] = cs->typesOFF[k];
      }
    }
    if (cs->mol_state[i] == 0) cs->mol_state[i] = 1;
    else if (cs->mol_state[i] == 1) cs->mol_state[i] = 0;
  }
  return 0;
}

int orc_cs_attempt_switch(orc_cluster_switch *cs, orc_atoms *a, const int *molecule)
{
  orc_cs_attempt_local(cs, a, molecule);
  return orc_cs_attempt_apply(cs, a);
}

/* a second object with the same parameters, survey and molecule arrays and its own random streams in the same state: what
   every rank of a decomposed run holds after the constructor's reductions (:114-116, :158-159; RanPark(lmp, seed) on every
   rank alike, :56-57) */
orc_cluster_switch *orc_cs_clone(const orc_cluster_switch *cs)
{
  orc_cluster_switch *c = (orc_cluster_switch *) malloc(sizeof(orc_cluster_switch));
  memcpy(c, cs, sizeof(orc_cluster_switch));
  const size_t nm = (size_t) cs->maxmol + 1;
#define ORC_DUP(field, count)                                        \
  do {                                                               \
    c->field = (int *) malloc(sizeof(int) * (size_t) ((count) > 0 ? (count) : 1)); \
    memcpy(c->field, cs->field, sizeof(int) * (size_t) (count));     \
  } while (0)
  ORC_DUP(typesON, cs->nSwitchTypes);
  ORC_DUP(typesOFF, cs->nSwitchTypes);
  ORC_DUP(contactMap, cs->nContactTypes * cs->nAtomsPerContact * 2);
  ORC_DUP(mol_restrict, nm);
  ORC_DUP(mol_state, nm);
  ORC_DUP(mol_accept, nm);
  ORC_DUP(mol_cluster, nm);
  ORC_DUP(mol_atoms, nm * (size_t) cs->nSwitchPerMol);
#undef ORC_DUP
  return c;
}

/* compute_vector (:887-897): attempts, successes, attempts ON/OFF, successes ON/OFF, cluster size */
void orc_cs_stats(const orc_cluster_switch *cs, double *out7)
{
  for (int k = 0; k < 6; k++) out7[k] = cs->stats[k];
  out7[6] = cs->nCluster;
}

int orc_cs_maxmol(const orc_cluster_switch *cs) { return cs->maxmol; }
const int *orc_cs_array(const orc_cluster_switch *cs, int which)
{
  return which == 0 ? cs->mol_cluster : which == 1 ? cs->mol_state : which == 2 ? cs->mol_restrict : cs->mol_accept;
}
