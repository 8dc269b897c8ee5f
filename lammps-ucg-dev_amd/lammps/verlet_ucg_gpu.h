/* -*- c++ -*- ----------------------------------------------------------
   USER-UCG/GPU: run_style verlet/ucg/gpu [comm rccl|mpi]

   The velocity-Verlet step loop of a deck whose per-step styles are all of this package -- one of the pair styles
   table_ucgld / table_ucg_bethe / table_ucg_bethe_density and the fixes nve/ucgld, nve/ucgld/wall/hard, ucgld/langevin,
   ucgstate, cluster_switch -- run INSIDE libucg_hip.so (ucg_md_setup / ucg_md_run_until, include/ucg_hip.h) with the whole
   state resident in HBM, on any number of ranks, one GPU each:

   * what upstream Verlet::setup() / run() do around the reference's styles -- exchange / borders /
     neighbour build on a re-neighbouring step, the forward communication of UCG/atom_vec_ucg.cpp:66-82's field lists on
     every other step, Neighbor::decide()'s all-reduce, the MPI_Allreduce steps of fix cluster_switch -- is done by the
     library on the device bricks `boxlo + prd * i / procgrid` of comm->procgrid (csrc/ucg_comm.hip);
   * between the ranks: RCCL called directly on the library's stream (default; rank 0's 128-byte id travels by MPI_Bcast) or,
     with `comm mpi`, four MPI callbacks on host-staged messages (ucg_comm_attach_host);
   * LAMMPS' per-atom arrays are brought up to date on output steps (Output::next), at the end of the run, and nowhere else:
     thermo, dump, restart, write_data and computes see the owned atoms of this rank's brick in the device's order.

   The fixes of the package carry their parameters to the device in init() and are not called per step.  Restrictions
   (checked in init()): orthogonal periodic box, uniform processor grid, no fix or per-step style from outside the package,
   all charges zero (atom style ucg carries q, the UCG styles do not use it and it does not travel with migrating atoms),
   image flags are not tracked (the device wraps positions into the box).

   Compiles only inside a LAMMPS source tree (needs integrate.h); see INTEGRATION.md.
------------------------------------------------------------------------- */
#ifdef INTEGRATE_CLASS
// clang-format off
IntegrateStyle(verlet/ucg/gpu,VerletUCGGPU);
// clang-format on
#else
#ifndef LMP_VERLET_UCG_GPU_H
#define LMP_VERLET_UCG_GPU_H

#include "integrate.h"

struct ucg_ctx;
struct ucg_pair;

namespace LAMMPS_NS {

class VerletUCGGPU : public Integrate {
 public:
  VerletUCGGPU(class LAMMPS *, int, char **);
  ~VerletUCGGPU() override;
  void init() override;
  void setup(int flag) override;
  void setup_minimal(int) override;
  void run(int) override;
  void force_clear();    // (pure virtual in Integrate since 2022; nothing to clear on the host)
  void cleanup() override;
  void reset_dt() override;

 private:
  ucg_ctx *ctx = nullptr;
  ucg_pair *gpair = nullptr;
  class PairTableUCGGPU *pair = nullptr;
  class FixUCGLDLangevinGPU *fix_lang = nullptr;
  class FixClusterSwitchGPU *fix_cs = nullptr;
  int use_nve = 0, use_lang = 0, use_ucgst = 0;
  int use_rccl = 1;          // `comm rccl` (default) | `comm mpi`
  bool attached = false;
  int me_grid = 0;           // this rank in the library's numbering: ix + px * (iy + py * iz)
  MPI_Comm gridworld;        // world re-ranked by me_grid (what the callbacks and the id broadcast use)
  bool have_gridworld = false;

  void check(int rc);
  void find_styles();
  void attach_communicator();
  void device_setup();
  void upload_atoms();
  void download_atoms(bool with_energy);

  // ucg_comm_ops callbacks over MPI (host-staged messages): csrc/ucg_comm.hip calls them between the device phases
  static int cb_alltoallv(void *user, const void *send, const long long *sendbytes, void *recv, const long long *recvbytes, void *);
  static int cb_alltoall_ll(void *user, const long long *send, long long *recv);
  static int cb_allreduce_ll(void *user, long long *buf, int n, int op);
  static int cb_allreduce_f64(void *user, double *buf, int n, int op);
};

}    // namespace LAMMPS_NS
#endif
#endif
