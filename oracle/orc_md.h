/* orc_md.h -- oracle MD driver state.  TEST INFRASTRUCTURE (see orc.h). */
#ifndef ORC_MD_H
#define ORC_MD_H

#include "orc.h"
#include "orc_cluster.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double boxlo[3], boxhi[3], prd[3];
  double sublo[3], subhi[3]; /* this rank's brick (= the box for a single-rank run) */
  double cutforce, skin, cutneigh;
  int every, delay, check; /* neigh_modify */
  int ntypes;
  double dt, boltz, ftm2v, mvv2e;
  int groupbit;
  int mode; /* 0 = reference order (half list + reverse sum), 1 = canonical gather */

  orc_atoms a;
  int nmax;
  double *xhold;
  int *ghost_src;   /* [nghost] owned index this ghost is an image of (on rank ghost_rank[g] in a decomposed run) */
  int *ghost_rank;  /* [nghost] decomposed runs (orc_world): the rank that owns the source bead, else unused */
  int *ghost_shift; /* [nghost*3] periodic shift in box lengths       */
  int *bin_of;      /* [nall] bin id at the last rebuild               */

  int nbin[3], sten[3], nbins;
  double bboxlo[3], binsize[3], bininv[3];
  int *binstart_owned, *binstart_ghost; /* [nbins+1] */

  orc_list full, half;

  orc_pair *pair;              /* borrowed */
  orc_fix_langevin *lang;      /* borrowed, may be NULL */
  orc_fix_ucgstate ucgst;
  int have_ucgstate, have_nve;  /* have_nve: 0 none, 1 nve/ucgld, 2 nve/ucgld/wall/hard, 3 the same with bias_potential */
  double wall_barrier;
  int *molecule;               /* [nall] molecule id (atom->molecule); default = tag */
  orc_cluster_switch *cs;      /* borrowed, may be NULL: fix cluster_switch */

  long long ntimestep, beginstep, endstep;
  int ago, nrebuild, pair_errors;
  orc_ev ev;
} orc_sim;

orc_sim *orc_sim_create(int natoms, const double *boxlo, const double *boxhi, double cutforce,
                        double skin, int ntypes);
void orc_sim_destroy(orc_sim *s);
void orc_sim_grow(orc_sim *s, int nmax);
orc_atoms *orc_sim_atoms(orc_sim *s);
orc_list *orc_sim_full_list(orc_sim *s);
orc_list *orc_sim_half_list(orc_sim *s);
void orc_sim_setup_bins(orc_sim *s);
void orc_sim_rebuild(orc_sim *s);
void orc_sim_forward_comm(orc_sim *s);
void orc_sim_reverse_comm(orc_sim *s);
int orc_sim_setup(orc_sim *s, long long nsteps_planned);
int orc_sim_run(orc_sim *s, long long nsteps, int thermo_every);
void orc_sim_set_run_params(orc_sim *s, double dt, int every, int delay, int check, int mode);
void orc_sim_set_units(orc_sim *s, double boltz, double ftm2v, double mvv2e);
void orc_sim_set_wall_barrier(orc_sim *s, double barrier);
int *orc_sim_molecule(orc_sim *s);
/* fix cluster_switch on this run (created from the sim's current atoms); returns NULL-error or message */
const char *orc_sim_cluster_switch(orc_sim *s, int mol_seed, int mol_offset, double cutoff, int seed, int switchFreq,
                                   const char *rateFile, const char *contactFile);
orc_cluster_switch *orc_sim_cs(orc_sim *s);
void orc_sim_attach(orc_sim *s, orc_pair *pair, orc_fix_langevin *lang, int have_nve,
                    int have_ucgstate, int ld_flag, int mc_flag, int mc_seed, double mc_rate);
void orc_sim_get_info(const orc_sim *s, long long *out);
void orc_sim_get_ev(const orc_sim *s, double *out);
const int *orc_sim_ghost_src(const orc_sim *s);
const int *orc_sim_ghost_shift(const orc_sim *s);
const int *orc_sim_bin_of(const orc_sim *s);
double *orc_sim_mass(orc_sim *s);
int orc_sim_compute_forces(orc_sim *s, int eflag, int vflag);

/* ---- decomposed runs: px x py x pz bricks, one orc_sim per rank (the per-rank semantics of a LAMMPS run with these
 * styles, as the GPU library's decomposed loop implements them -- csrc/ucg_comm.hip, "ucg-rebuild-v1" per brick):
 *   rank me = ix + px (iy + py iz) owns the beads whose wrapped position lies in its brick [boxlo + prd i / p, ...);
 *   a re-neighbouring wraps, migrates every bead to its owner, sorts each rank's beads by (Morton bin of the brick's
 *   own bin grid, tag), makes rank r's ghosts from every (bead of any rank, box shift) image inside r's brick extended
 *   by the list cutoff -- its own unshifted beads excepted -- sorted by (Morton bin, tag, shift code), and builds r's rows;
 *   every step the ghosts take x + shift, lambda, ucgp, state from their owners (forward halo; no reverse halo in the
 *   canonical gather order); Neighbor::decide is the MAX of the ranks' flags (upstream MPI_Allreduce);
 *   the random streams are per rank: RanMars(seed + me) drawn in the rank's local bead order
 *   (UCG/fix_ucgld_langevin.cpp:85, 280; UCG/fix_ucgstate.cpp:62, 117).
 * All three pair styles in canonical order (the density style's passes run in lockstep over the ranks, its ghosts' priors
 * and CV forces coming from their owner ranks); fix cluster_switch with the reference's reductions between the ranks' label
 * sweeps and decisions, one RanPark stream per rank (orc_world_cluster_switch). */
typedef struct orc_world orc_world;
orc_world *orc_world_create(const int *grid3, int natoms, const double *boxlo, const double *boxhi, double cutforce,
                            double skin, int ntypes);
void orc_world_destroy(orc_world *w);
int orc_world_nranks(const orc_world *w);
orc_sim *orc_world_rank(orc_world *w, int r);
/* natoms beads handed to rank 0 as a start (the first re-neighbouring sends every bead to its owner) */
orc_atoms *orc_world_input(orc_world *w);
int *orc_world_input_molecule(orc_world *w);
void orc_world_set_run_params(orc_world *w, double dt, int every, int delay, int check);
/* fixes on every rank: langevin (NULL-able by have_langevin = 0) with seed + me, ucgstate with mc_seed + me */
void orc_world_attach(orc_world *w, orc_pair *pair, int have_langevin, double t_start, double t_stop, double t_period,
                      int lang_seed, int have_nve, double wall_barrier, int have_ucgstate, int ld_flag, int mc_flag,
                      int mc_seed, double mc_rate);
/* fix cluster_switch on every rank (survey reduced over the ranks, RanPark streams per rank, seeded alike); call it before
 * orc_world_setup.  Returns NULL or the error message */
const char *orc_world_cluster_switch(orc_world *w, int mol_seed, int mol_offset, double cutoff, int seed, int switchFreq,
                                     const char *rateFile, const char *contactFile);
int orc_world_setup(orc_world *w, long long nsteps_planned);
int orc_world_run(orc_world *w, long long nsteps, int thermo_every);
/* totals over the ranks of the last energy evaluation: eng_vdwl, virial[6] */
void orc_world_get_ev(const orc_world *w, double *out7);

#ifdef __cplusplus
}
#endif
#endif
