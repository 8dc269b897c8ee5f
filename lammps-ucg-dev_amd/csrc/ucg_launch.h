// ucg_launch.h -- host-callable launchers of the HIP kernels (internal to libucg_hip).
#pragma once

#include <hip/hip_runtime.h>

#include "ucg_dev.h"

#define UCG_MAX_ACTUAL 7
#define UCG_MAX_TABLES 64

namespace ucg {

// ---- ucg_pair.hip
int pair_gather_blocks(int nlocal, int slots);
hipError_t launch_pair_gather(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev,
                              double *evpart, double *evout, int *errflag, hipStream_t st);
hipError_t launch_pair_gather_fused(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev,
                              double *evpart, double *evout, int *errflag, hipStream_t st);
hipError_t launch_block_classify(const AtomsDev &A, const ListDev &L, int slots, int *flags, hipStream_t st);

hipError_t launch_ev_final(const double *part, int nblocks, double *out, hipStream_t st);

// ---- ucg_pair_vrow.hip: own-block pairs once on balanced virtual rows, fixed sums
int vrow_blocks(int nlocal);
int vrow_beads();
int vrow_maxrow();
int vrow_lists();
int vrow_capacity(int maxrow);
size_t vrow_lds_bytes(const PairDev &P);
hipError_t launch_vrow_build(const PairDev &P, const AtomsDev &A, const ListDev &L, double skin, int *ent, size_t list_stride,
                             int cap, int vpitch, int2 *lanemeta, int *errflag, hipStream_t st);
hipError_t launch_pair_vrow(const PairDev &P, const AtomsDev &A, const ListDev &L, const int *ent, size_t list_stride,
                            const int2 *lanemeta, int vpitch, bool ev, double *evpart, double *evout, int *errflag,
                            hipStream_t st);

hipError_t launch_selftest_div(double b, unsigned long long seed, int n, unsigned long long *d_mismatches, hipStream_t st);
hipError_t launch_selftest_div_core(unsigned long long seed, int n, unsigned long long *d_mismatches, hipStream_t st);
hipError_t launch_selftest_sqrt_core(unsigned long long seed, int n, unsigned long long *d_mismatches, hipStream_t st);

// ---- ucg_density.hip
hipError_t launch_density(const PairDev &P, const AtomsDev &A, const ListDev &L, const int *ghost_src, bool ev,
                          double2 *prior, double *partial0, double2 *cv, double *evpart, double *evout, int *errflag,
                          hipStream_t st);
hipError_t launch_density_phase(const PairDev &P, const AtomsDev &A, const ListDev &L, int phase, bool ev, double2 *prior,
                                double *partial0, double2 *cv, double *evpart, double *evout, int *errflag,
                                hipStream_t st);
int density_evpart_doubles(int nlocal);

// ---- ucg_fix.hip
struct LangevinDev {
  const double *gfactor1, *gfactor2;  // [ntypes+1]
  double tsqrt;
  const unsigned int *draws;  // 24-bit RanMars integers, one per owned bead
  int bias;                   // post_force_templated<1>: no random force on a bead whose lambda velocity is exactly 0
};
hipError_t launch_nve_initial(const AtomsDev &A, double dtv, double dtf, int groupbit, int wall, hipStream_t st);
hipError_t launch_nve_final(const AtomsDev &A, double dtf, int groupbit, int wall, hipStream_t st);
hipError_t launch_wall_bias(const AtomsDev &A, double barrier, int groupbit, hipStream_t st);
hipError_t launch_langevin(const AtomsDev &A, const LangevinDev &Lg, int groupbit, hipStream_t st);
hipError_t launch_lambda_ke(const AtomsDev &A, int groupbit, double mvv2e, double *part, double *out,
                            hipStream_t st);
hipError_t launch_ucgstate(const AtomsDev &A, int ld_flag, int mc_flag, double mc_rate,
                           const unsigned int *draws, hipStream_t st);
hipError_t launch_force_clear(const AtomsDev &A, hipStream_t st);
hipError_t launch_post_fused(const AtomsDev &A, bool lang, const LangevinDev &Lg, bool ucgst, int ld_flag, int mc_flag,
                             double mc_rate, const unsigned int *mc_draws, bool nve, bool next, double dtv, double dtf,
                             int groupbit, int wall, double barrier, hipStream_t st);
hipError_t launch_stream(const void *buf, size_t nbytes, int wide, int *sink, hipStream_t st);

// ---- ucg_ranmars.hip : exact block-parallel RANMAR
struct RanMarsDev {
  // two history buffers (ping-pong), 97 lag values each, oldest first, 24-bit integers
  unsigned int *hist[2];
  int cur;
  long long count;           // uniform() calls made so far, constructor warm-up included
  const unsigned int *jump;  // [nchunks_max][97] coefficients of z^(p*CHUNK) mod P(z)
  int nchunks_max;
};
constexpr int RANMARS_CHUNK = 1024;
// host: RANMAR seeding (LAMMPS convention) -> 97 lag values oldest first after the warm-up draw
void ranmars_seed_host(int seed, unsigned int *hist97, long long *count);
// host: jump polynomials z^(p*CHUNK) mod (z^97 + z^64 - 1) over Z/2^24, p = 0..nchunks-1
void ranmars_jump_host(int nchunks, unsigned int *out);
// n consecutive draws -> out[0..n) as 24-bit integers (uniform = out * 2^-24); advances R
hipError_t launch_ranmars(RanMarsDev &R, int n, unsigned int *out, hipStream_t st);

}  // namespace ucg
