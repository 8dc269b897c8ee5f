/* orc.h -- CPU oracle for the UCG hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (lammps-ucg-dev_amd/, include/) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * PARITY UNPINNED: the reference (KJAdams2000/LAMMPS-UCG-dev) ships no tests,
 * golden vectors or fixtures for this path, and it cannot be compiled here
 * (it needs upstream LAMMPS headers that are absent; writing stand-ins for
 * them is not allowed).  This oracle is a line-by-line restatement of the
 * reference's arithmetic, each function citing the reference file:line it
 * follows.  Pieces of upstream LAMMPS the path relies on (RanMars, RanPark,
 * ev_tally, Verlet ordering, table file reader) are restated from their
 * published algorithms; RanMars is pinned by the published RANMAR check
 * values (tests/test_oracle.py).
 *
 * Compile with -O2 -ffp-contract=off (no FMA fusion: the reference is plain
 * x86-64 code, and the HIP kernels are built the same way).
 */
#ifndef ORC_H
#define ORC_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ RNG */

typedef struct {
  double u[98];
  int i97, j97;
  double c, cd, cm;
} orc_ranmars;

void orc_ranmars_init(orc_ranmars *r, int seed);
double orc_ranmars_uniform(orc_ranmars *r);
/* n draws into out[] (convenience for tests) */
void orc_ranmars_fill(orc_ranmars *r, int n, double *out);

typedef struct { int seed; } orc_ranpark;
void orc_ranpark_init(orc_ranpark *r, int seed);
double orc_ranpark_uniform(orc_ranpark *r);

/* math selection: 0 = the oracle's own fdlibm-shaped functions (orc_math.c: the written definition
 *                     "ucg-math-v1", bit-reproducible; nothing of the product tree is included),
 *                 1 = libm (what the reference itself calls)             */
void orc_set_math(int use_libm);
int orc_get_math(void);
double orc_exp(double x);
double orc_expm1(double x);
double orc_log(double x);
double orc_tanh(double x);

/* --------------------------------------------------------------- tables */

enum { ORC_LOOKUP = 0, ORC_LINEAR = 1, ORC_SPLINE = 2, ORC_BITMAP = 3 };
enum { ORC_RNONE = 0, ORC_RLINEAR = 1, ORC_RSQ = 2, ORC_BMP = 3 };

typedef struct {
  int ninput, rflag, fpflag, match;
  double rlo, rhi, fplo, fphi, cut;
  double *rfile, *efile, *ffile, *e2file, *f2file;
  double innersq, delta, invdelta, deltasq6;
  double *rsq, *e, *f, *de, *df, *e2, *f2;
  /* BITMAP tables (:1247-1340): 2^tablength bins addressed by the bits of (float) rsq */
  double *drsq;
  int ntablebits, nmask, nshiftbits;
} orc_table;

/* upstream Pair::init_bitmap (src/pair.cpp, LAMMPS stable releases; not part of the reference tree): which bits of
 * a float in [inner^2, outer^2] index a 2^ntablebits table.  Returns 0, or 1 with a message. */
int orc_init_bitmap(double inner, double outer, int ntablebits, int *masklo, int *maskhi, int *nmask,
                    int *nshiftbits, char *err, int errlen);

void orc_spline(const double *x, const double *y, int n, double yp1, double ypn, double *y2);
double orc_splint(const double *xa, const double *ya, const double *y2a, int n, double x);

/* returns 0 on success; on failure writes a message to err (len errlen) */
int orc_table_read(orc_table *tb, const char *file, const char *keyword, char *err, int errlen);
int orc_table_from_arrays(orc_table *tb, int ninput, const double *r, const double *e,
                          const double *f, int rflag, double rlo, double rhi, int fpflag,
                          double fplo, double fphi);
int orc_table_build(orc_table *tb, int tabstyle, int tablength, double cut, char *err, int errlen);
void orc_table_free(orc_table *tb);
/* (f/r, e) from one table, the open-coded block of the reference's inner loop */
int orc_table_eval(const orc_table *tb, int tabstyle, int tablength, double rsq, double *fval,
                   double *eval);

/* ----------------------------------------------------------- pair model */

enum { ORC_STYLE_UCGLD = 0, ORC_STYLE_BETHE = 1, ORC_STYLE_BETHE_DENSITY = 2 };
enum { ORC_PRIOR_CHEMPOT = 0, ORC_PRIOR_CHEMPOT_NOISE = 1, ORC_PRIOR_UCGL = 2, ORC_PRIOR_UCGP = 3 };
enum { ORC_METHOD_MF = 0, ORC_METHOD_BETHE = 1 };

typedef struct {
  int style;
  int tabstyle, tablength;
  int n_actual, n_formal, max_states;
  int *n_states_per_type;    /* [n_actual+1]              */
  int *formal_from_actual;   /* [(n_actual+1)*max_states] */
  int *actual_from_formal;   /* [n_formal+1]              */
  double *chem_pot;          /* [n_formal+1]              */
  int ntables;
  orc_table *tables;
  int allocated;
  int *tabindex, *setflag;   /* [(n_formal+1)^2] */
  double *cutsq;             /* [(n_formal+1)^2] */
  double special_lj[4];
  double T, kT;
  /* table_ucg_bethe options (UCG/pair_table_ucg_bethe.cpp:756-759) */
  int pseudo_flag, prior_flag, method_flag, seed;
  double noise_level;
  orc_ranmars random;
  int have_random;
  double *prior_prob_from_type; /* [(n_actual+1)*max_states] */
  /* table_ucg_bethe_density options */
  int *use_density, *use_state_entropy; /* [n_actual+1] */
  double *cv_thresholds, *threshold_radii; /* [n_actual+1] */
  /* density-style compat switches; 0 = fixed, 1 = as shipped */
  int density_proximity_as_shipped;  /* #12 */
  /* canonical gather order: number of interleaved slot accumulators per bead (power of two, <= ORC_MAX_SLOTS) */
  int gather_slots;
  /* "fixed" sums (the library's pair kernels that evaluate a pair of one workgroup once, csrc/ucg_pair_vrow.hip):
     sum_fixed != 0 -> a bead's terms are rounded to nearest-even at 2^-38 of a per-field power-of-two unit
     (2^-sum_exp[field]; field 0 force components, 1 ucgforce, 2 scores) and added as 64-bit integers: associative and
     commutative, so no order is part of the definition.  bead total = prologue value + (double) integer sum * unit.
     Sets *err bit 4 when a term's image does not fit. */
  int sum_fixed;
  int sum_exp[3];
  char errmsg[512];
} orc_pair;

#define ORC_MAX_SLOTS 64
void orc_pair_set_sum_fixed(orc_pair *p, int on);
void orc_pair_sum_scales(orc_pair *p); /* called by orc_pair_init */
orc_pair *orc_pair_create(int style);
void orc_pair_destroy(orc_pair *p);
const char *orc_pair_error(const orc_pair *p);
/* same argument lists as the LAMMPS commands (after the style name) */
int orc_pair_settings(orc_pair *p, int narg, const char **arg);
int orc_pair_coeff(orc_pair *p, int ntypes, int narg, const char **arg);
/* Pair::init(): init_style + init_one for all i<=j<=ntypes; T = thermostat t_target */
int orc_pair_init(orc_pair *p, int ntypes, double T, double boltz);

/* ----------------------------------------------------- atoms and lists */

typedef struct {
  int nlocal, nghost;
  double *x;      /* [nall*3] */
  double *v;      /* [nlocal*3] */
  double *f;      /* [nall*3] */
  int *type, *tag, *mask;
  int *ucgstate, *num_ucgstates;
  double *ucgl, *ucgvl, *ucgml, *ucgp, *ucgforce;
  double *scores; /* ucgsoftmaxscores [nall*2] */
  double *mass;   /* [ntypes+1] */
} orc_atoms;

/* CSR neighbour list.  Entry bits: [28:0] index, [29] orientation (set when
 * the row owner plays the reference's "i" role for this pair), [31:30] the
 * LAMMPS special-bond code (sbmask).  Half lists ignore bit 29.            */
#define ORC_NEIGHMASK 0x1FFFFFFF
#define ORC_ORIENT_BIT 29
#define ORC_SBBITS 30
typedef struct {
  int inum;
  int *ilist;     /* [inum] */
  int *numneigh;  /* [inum] indexed by ii */
  long long *first; /* [inum] offset of row ii in neigh */
  int *neigh;
} orc_list;

typedef struct {
  double eng_vdwl;
  double virial[6];
  int err;          /* 0 ok, 1 "< inner cutoff", 2 "> outer cutoff" */
  int err_i, err_j; /* first offending pair */
} orc_ev;

/* reference order: sequential half list with scatter to j (newton on) */
int orc_pair_compute_half(orc_pair *p, orc_atoms *a, const orc_list *l, int newton_pair,
                          int eflag, int vflag, orc_ev *ev);
/* canonical order: per owned atom gather over a full list, in list order */
int orc_pair_compute_gather(orc_pair *p, orc_atoms *a, const orc_list *l, int eflag, int vflag,
                            orc_ev *ev);

/* table_ucg_bethe_density (orc_density.c): mode 0 = sequential sweep with scatter (the
 * reference's loop shape), 1 = canonical gather; ghost_src[g] = owned index ghost g images */
/* the same in three calls on shared work arrays [nall][2] (fpart [nall][3]): a decomposed run moves the ghosts' entries of
 * prior / partial (after pass 1) and cv (after pass 2) between ranks itself */
typedef struct {
  int nall;
  double *prior, *partial, *G, *S, *cv, *fpart;
} orc_density_work;
orc_density_work *orc_density_work_create(int nall);
void orc_density_work_destroy(orc_density_work *w);
int orc_pair_density_check(orc_pair *p, const orc_atoms *a);
void orc_pair_density_pass1(orc_pair *p, orc_atoms *a, const orc_list *l, orc_density_work *w);
void orc_pair_density_pass2(orc_pair *p, orc_atoms *a, const orc_list *l, int mode, int eflag, int vflag, orc_density_work *w,
                            orc_ev *ev);
void orc_pair_density_pass3(orc_pair *p, orc_atoms *a, const orc_list *l, int mode, int vflag, const int *ghost_src,
                            orc_density_work *w, orc_ev *ev);
int orc_pair_density_compute(orc_pair *p, orc_atoms *a, const orc_list *l, int mode, int eflag,
                             int vflag, const int *ghost_src, orc_ev *ev);

/* --------------------------------------------------------------- fixes */

/* AtomVecUCG::force_clear + Verlet::force_clear (UCG/atom_vec_ucg.cpp:131-135) */
void orc_force_clear(orc_atoms *a, int include_ghosts);

/* fix nve/ucgld (UCG/fix_nve_ucgld.cpp:36-153) */
void orc_fix_nve_initial(orc_atoms *a, double dt, double ftm2v, int groupbit);
void orc_fix_nve_final(orc_atoms *a, double dt, double ftm2v, int groupbit);
/* fix nve/ucgld/wall/hard (UCG/fix_nve_ucgld_wall_hard.cpp:61-241) */
void orc_fix_nve_wall_initial(orc_atoms *a, double dt, double ftm2v, int groupbit);
void orc_fix_nve_wall_final(orc_atoms *a, double dt, double ftm2v, int groupbit);
double orc_wall_bias_force(double lmd, double H);
void orc_fix_nve_wall_post_force(orc_atoms *a, double barrier, int groupbit);

/* fix ucgld/langevin (UCG/fix_ucgld_langevin.cpp) */
typedef struct {
  double t_start, t_stop, t_period, t_target, tsqrt;
  int seed, ntypes;
  double *gfactor1, *gfactor2; /* [ntypes+1] */
  orc_ranmars random;
  double lambda_temp;
} orc_fix_langevin;

orc_fix_langevin *orc_fix_langevin_create(int ntypes, double t_start, double t_stop,
                                          double t_period, int seed, int me);
void orc_fix_langevin_destroy(orc_fix_langevin *fx);
/* init(): :149-183; reads a->ucgml[type index] exactly like the reference (App. B #5) */
void orc_fix_langevin_init(orc_fix_langevin *fx, const orc_atoms *a, double dt, double boltz,
                           double ftm2v, double mvv2e);
void orc_fix_langevin_post_force(orc_fix_langevin *fx, orc_atoms *a, int groupbit,
                                 long long ntimestep, long long beginstep, long long endstep);
void orc_fix_langevin_end_of_step(orc_fix_langevin *fx, const orc_atoms *a, int groupbit,
                                  double boltz, double mvv2e);

/* fix ucgstate (UCG/fix_ucgstate.cpp:88-132) */
typedef struct {
  int ld_flag, mc_flag, mc_seed;
  double mc_rate;
  orc_ranmars random;
} orc_fix_ucgstate;
void orc_fix_ucgstate_init(orc_fix_ucgstate *fx, int ld_flag, int mc_flag, int mc_seed,
                           double mc_rate, int me);
void orc_fix_ucgstate_post_force(orc_fix_ucgstate *fx, orc_atoms *a);

#ifdef __cplusplus
}
#endif
#endif
