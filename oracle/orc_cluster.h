/* orc_cluster.h -- fix cluster_switch restated for the oracle.  TEST INFRASTRUCTURE (see orc.h). */
#ifndef ORC_CLUSTER_H
#define ORC_CLUSTER_H

#include "orc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int mol_seed, mol_offset, switchFreq, groupbit;
  double cutsq;
  orc_ranpark random_equal, random_unequal;
  double probON, probOFF;
  int nSwitchTypes, *typesON, *typesOFF;
  int nContactTypes, nAtomsPerContact, *contactMap; /* [nContactTypes][nAtomsPerContact][2] */
  int maxmol, nmol, nSwitchPerMol;
  int *mol_restrict, *mol_state, *mol_accept, *mol_cluster, *mol_atoms;
  long long next_reneighbor;
  double stats[6], nCluster;
  int sweeps; /* sweeps of the last check_cluster (diagnostic) */
  char errmsg[256];
} orc_cluster_switch;

/* fix ID group cluster_switch mol_seed mol_offset cutoff seed rateFreq N rateFile F contactFile F
 * (UCG/fix_cluster_switch.cpp:37-181); on an input error the object is returned with orc_cs_error() set */
orc_cluster_switch *orc_cs_create(const orc_atoms *a, const int *molecule, int ntypes, int groupbit, int mol_seed,
                                  int mol_offset, double cutoff, int seed, int switchFreq, const char *rateFile,
                                  const char *contactFile, long long ntimestep);
void orc_cs_destroy(orc_cluster_switch *cs);
const char *orc_cs_error(const orc_cluster_switch *cs);
/* molecule[] covers owned + ghost atoms; list is a full list */
int orc_cs_check_cluster(orc_cluster_switch *cs, const orc_atoms *a, const int *molecule, const orc_list *list);
int orc_cs_attempt_switch(orc_cluster_switch *cs, orc_atoms *a, const int *molecule);
/* the phases of the two, for decomposed runs (orc_world reduces over the ranks in between, as the reference's MPI_Allreduce
 * calls do): presence / labels_init / sweep_local / finalize, attempt_local / attempt_apply; and a per-rank copy */
void orc_cs_presence(const orc_cluster_switch *cs, const orc_atoms *a, const int *molecule, int *present);
void orc_cs_labels_init(orc_cluster_switch *cs, const int *present, int *lab);
int orc_cs_sweep_local(orc_cluster_switch *cs, const orc_atoms *a, const int *molecule, const orc_list *list, int *lab);
void orc_cs_finalize(orc_cluster_switch *cs, const int *lab);
void orc_cs_attempt_local(orc_cluster_switch *cs, const orc_atoms *a, const int *molecule);
int orc_cs_attempt_apply(orc_cluster_switch *cs, orc_atoms *a);
orc_cluster_switch *orc_cs_clone(const orc_cluster_switch *cs);
void orc_cs_stats(const orc_cluster_switch *cs, double *out7);
int orc_cs_maxmol(const orc_cluster_switch *cs);
/* which: 0 mol_cluster, 1 mol_state, 2 mol_restrict, 3 mol_accept; maxmol+1 entries */
const int *orc_cs_array(const orc_cluster_switch *cs, int which);

#ifdef __cplusplus
}
#endif
#endif
