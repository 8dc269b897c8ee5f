// ucg_dev.h -- device-side layouts shared by the HIP kernels (gfx950).
//
// Data layout in HBM (all fp64 / int32, DESIGN.md section "HBM layout"):
//   pos4[nall]   double4 {x, y, z, lambda}   -- lambda (ucgl) rides as the 4th coordinate
//   vel4[nlocal] double4 {vx, vy, vz, vlambda}
//   frc4[nlocal] double4 {fx, fy, fz, ucgforce}
//   scores[nlocal] double2 ucgsoftmaxscores
//   meta[nall]   int32  type | ucgstate << 16
//   ucgp[nall]   double
//   tag[nall], mask[nlocal], num_ucgstates[nlocal] int32; ucgml[nlocal] double
//   neigh[maxrow][pitch] int32, row-transposed full list (entry e of bead k at e*pitch+k),
//     so that a wavefront reads 64 consecutive ints per entry slot.
//   tables: double4 per knot  SPLINE {e, f, e2, f2} | LINEAR {e, de, f, df} | LOOKUP {e, f, 0, 0}
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define UCG_META_TYPE(m) ((m) & 0xFFFF)
#define UCG_META_STATE(m) (((m) >> 16) & 1)

namespace ucg {

// Kind blocks (table_ucg_bethe_density's pass 2 on a deck of several actual types whose tables are read through L1 / L2):
// the tables of every (row type a, neighbour type b) kind as one compact FAST block -- per knot {t00, t01, [t10,] t11} x two
// 16-byte slots + one padding slot -- so that what a cold lane reads for one pair lies in two or three cache lines instead
// of being spread over the full layout's 2 * ntab + 1 slots per knot.  Same values either way.
struct KindsDev {
  const double4 *kind_tab;  // all blocks; nullptr: none
  const int2 *kind_dir;     // [(n_actual+1)^2] {offset into kind_tab in double4 units, tables of the kind: 3 or 4}
};

struct PairDev {
  int style, tabstyle, tablength, tlm1;
  int n_actual;        // actual types are 1..n_actual
  int onetype_same10;  // n_actual == 1 and tables (0,1) and (1,0) of the pair (1,1) are one table: the ONETYPE kernels apply
  int ntab;            // tables resident on the device (those reachable through tabindex)
  int tab_in_lds;      // 1: the kernels stage all tables in LDS
  int pseudo_flag, prior_flag, method_flag;
  int first_possible;  // table_ucg_bethe: some bead may still carry the first-call marker ucgp < -0.999 (ucg_ctx::ucgp_first_possible)
  const double4 *tab;     // [ntab * tablength]
  const double4 *tab_fast;  // FAST layout: [tablength][2*ntab+1] 16-byte slots (see ucg_pair.hip)
  int fast_stride;        // 2*ntab+1
  const double4 *tabpar;  // [ntab] {innersq, delta, invdelta, deltasq6}
  const int *pairtab;     // [(n_actual+1)^2 * 4] table of (ti, tj, a, b)
  const double *cutsq;    // [(n_actual+1)^2]  cutsq[itype][jtype] as the reference indexes it
  const double *mu;       // [(n_actual+1)*2]  chemical potential of state s of actual type t
  const double *prior_type;  // [(n_actual+1)*2]
  // table_ucg_bethe_density, per actual type
  const int *dens_flags;    // [(n_actual+1)*2] {use_density, use_state_entropy}
  const double *dens_par;   // [(n_actual+1)*2] {cv_threshold, threshold_radius}
  int dens_as_shipped;      // 1: back-force uses the proximity function itself (App. B #12)
  double kT;
  double rkT;          // RN(1/kT), used by the FAST kernels' exact division
  int kT_pow2;         // kT is a power of two: a * rkT is the exact quotient
  int gather_slots;    // lanes per bead in k_pair_gather (1, 4, 8 or 16): part of the canonical order
  int stage_own;       // 1: k_pair_gather keeps its workgroup's own beads in LDS behind the tables
  int stage_own_allowed;  // the context option "stage_own" (kernels with other block shapes decide the fit themselves)
  // fixed sums (ucg_pair_dev.h; set when the pair runs on virtual rows): power-of-two units of the force / ucgforce /
  // score sums, their inverses * 2^-38, and the r^2 beyond which every term of a pair is below 8192 units for
  // |lambda - 0.5| <= 2.3
  double sum_sc[3], sum_dec[3], sum_rsq_safe;
  int fast;            // 1: one shared r^2 grid, all special_lj == 1, kT usable for div_by_const
  // FAST tables that do not fit the LDS (several actual types: read through L1 / L2): the three tables of the pairs of
  // ONE actual type with itself -- the most populous one, chosen by the host -- are staged in LDS all the same, and a lane
  // whose pair is of that kind reads them there (per-lane generic pointers); 0 = none
  int hot_type;
  int hot_ent;            // double4 entries of the hot block: (tablength * 7 + 1) / 2 (the one-type FAST layout, stride 7)
  // ... or, ONE actual type whose tables are too long for the LDS: hot_k0 >= 0 and the LDS holds the knots hot_k0 ..
  // tablength - 1 of the FAST layout as it is (the far end of the r^2 grid, where most pairs are); a lane whose knot
  // lies in that window reads it there
  int hot_k0;
  const double4 *tab_hot;
  double special_lj[4];
  KindsDev kinds;
  // table_ucg_bethe_density: tanh of the proximity argument of every in-cutoff entry, written by pass 1 at the entry's place
  // (e * pitch + k) and read back by pass 3, which needs the same value for the same pair (nullptr: pass 3 evaluates it again)
  double *tcache;
};

struct AtomsDev {
  int nlocal, nghost;
  double4 *pos4;
  double4 *vel4;
  double4 *frc4;
  double2 *scores;
  int *meta;
  double *ucgp;
  int *tag;
  int *mask;
  int *num_ucgstates;
  double *ucgml;
  const double *mass;  // [ntypes+1]
};

// Optional epilogue of the gather kernels: the per-bead hooks that follow the pair force on a step whose
// next initial_integrate is fused in (k_post_fused<.., NEXT = true>), done by the lane that has just summed
// the bead's forces -- they never travel through HBM.  Positions / states of the next step go to a second
// buffer (other workgroups are still reading the current ones); the host swaps the buffers afterwards.
struct PostDev {
  int enabled;
  int lang, ucgst, nve;  // which hooks (nve: 0 none, 1 nve/ucgld, 2 nve/ucgld/wall/hard, 3 + bias_potential)
  int ld_flag, mc_flag, groupbit;
  int lang_bias;         // LangevinDev::bias
  double mc_rate, tsqrt, dtv, dtf, barrier;
  const double *gfactor1, *gfactor2;
  const unsigned int *lang_draws, *mc_draws;
  double4 *pos_out;
  int *meta_out;
  double *ucgp_out;  // written (and swapped in by the host) only when ucgst is set
};

struct ListDev {
  int inum;
  int pitch;
  int maxrow;
  const int *neigh;
  const int *numneigh;
  // optional workgroup filter of the gather kernels (decomposed runs overlap the halo with the
  // workgroups that touch no ghost): run only workgroups with blockflag[chunk] == blockwant
  const int *blockflag;
  int blockwant;
  // 1: the rows are larger than the last-level cache (ucg_ctx::list_dev: more than 192 MB of entries) and are read with
  // non-temporal loads, so that they do not displace the beads the gathers re-read; smaller lists stay cached from step to step
  int stream_rows;
  PostDev post;
};

}  // namespace ucg
