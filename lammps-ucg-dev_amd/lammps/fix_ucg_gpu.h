/* -*- c++ -*- ----------------------------------------------------------
   USER-UCG/GPU: the UCG fixes backed by libucg_hip.so (include/ucg_hip.h), with the
   reference's style names and argument lists:

     fix ID group nve/ucgld                                  (UCG/fix_nve_ucgld.h:15-16)
     fix ID group nve/ucgld/wall/hard [bias_potential [H]]   (UCG/fix_nve_ucgld_wall_hard.h:12)
     fix ID group ucgld/langevin Tstart Tstop damp seed      (UCG/fix_ucgld_langevin.h:14-17)
     fix ID group ucgstate [ld | mc seed rate]               (UCG/fix_ucgstate.h:1-3)
     fix ID group cluster_switch molSeed molOffset cutoff seed rateFreq N rateFile F contactFile F
                                                             (UCG/fix_cluster_switch.cpp:37-60)

   The device context is the pair style's (Pair::extract("ucg_ctx")).  Under `run_style verlet/ucg/gpu` (verlet_ucg_gpu.h) the
   whole step loop runs inside the library -- any number of ranks, RCCL or MPI between them -- and the hooks below are not
   called at all: the fixes only carry their parameters to the device in init().  Under the stock `run_style verlet` there
   are two modes, chosen by the pair style in init_style():
   * resident (one rank, every fix of the deck is one of these): the device arrays are authoritative between the hooks and
     LAMMPS' arrays are mirrors bound with ucg_host_bind; a hook launches its kernel and moves nothing.  The integrator
     takes Neighbor::decide() over (distance check on the device, re-neighbouring forced through force_reneighbor /
     next_reneighbor), synchronises what exchange / borders read in pre_exchange() and everything on output steps;
   * copy mode (anything else): every hook refreshes what it reads (ucg_atoms_upload_owned), launches the kernel and
     copies back what it wrote (ucg_atoms_download).

   Compiles only inside a LAMMPS source tree (needs fix.h); see INTEGRATION.md.
------------------------------------------------------------------------- */
#ifdef FIX_CLASS
// clang-format off
FixStyle(nve/ucgld,FixNVEUCGLDGPU);
FixStyle(nve/ucgld/wall/hard,FixNVEUCGLDWallHardGPU);
FixStyle(ucgld/langevin,FixUCGLDLangevinGPU);
FixStyle(ucgstate,FixUCGStateGPU);
FixStyle(cluster_switch,FixClusterSwitchGPU);
// clang-format on
#else
#ifndef LMP_FIX_UCG_GPU_H
#define LMP_FIX_UCG_GPU_H

#include "fix.h"

#include <string>

struct ucg_ctx;

namespace LAMMPS_NS {

// shared plumbing: find the pair style's device context, move fields either way
class FixUCGGPUBase : public Fix {
 public:
  FixUCGGPUBase(class LAMMPS *, int, char **);
  void init() override;

 protected:
  ucg_ctx *ctx = nullptr;
  int resident = 0;    // the pair style's extract("ucg_resident")
  int driver = 0;      // the pair style's extract("ucg_driver"): run_style verlet/ucg/gpu owns the step loop
  void check(int rc);
  enum { X = 1, V = 2, F = 4, STATE = 8, NSTATES = 16, L = 32, VL = 64, P = 128, LF = 256, SCORES = 512 };
  void to_device(int fields);
  void from_device(int fields);
};

class FixNVEUCGLDGPU : public FixUCGGPUBase {
 public:
  FixNVEUCGLDGPU(class LAMMPS *, int, char **);
  ~FixNVEUCGLDGPU() override;
  int setmask() override;
  void init() override;
  void initial_integrate(int) override;
  void final_integrate() override;
  void pre_exchange() override;
  void end_of_step() override;
  void initial_integrate_respa(int, int, int) override;
  void final_integrate_respa(int, int) override;
  void post_run() override;
  void reset_dt() override;
  bool is_wall() const { return wall; }

 protected:
  bool wall = false;
  double *step_respa = nullptr;
  void set_step(double dt);
  // resident mode takes Neighbor::decide() over for the duration of ONE run: the user's neigh_modify values are put back in
  // post_run() (and by the destructor), so the next `run` -- resident or not -- starts from what the input deck said
  bool neigh_taken = false;
  int saved_delay = 0, saved_every = 1;
  void restore_neigh_modify();
};

class FixNVEUCGLDWallHardGPU : public FixNVEUCGLDGPU {
 public:
  FixNVEUCGLDWallHardGPU(class LAMMPS *, int, char **);
  int setmask() override;
  void init() override;
  void post_force(int) override;
  int has_bias_potential() const { return bias_potential_flag; }

 protected:
  int bias_potential_flag = 0;
  double barrier = 0.1;
};

class FixUCGLDLangevinGPU : public FixUCGGPUBase {
 public:
  FixUCGLDLangevinGPU(class LAMMPS *, int, char **);
  int setmask() override;
  void init() override;
  void setup(int) override;
  void post_force(int) override;
  void post_force_respa(int, int, int) override;
  void end_of_step() override;
  void reset_target(double) override;
  void reset_dt() override;
  int modify_param(int, char **) override;
  double compute_scalar() override;
  void *extract(const char *, int &) override;
  ~FixUCGLDLangevinGPU() override;
  // run_style verlet/ucg/gpu reports the values of the last output step here (the hooks above are not called under it)
  void set_from_driver(double t_target_now, double lambda_temp_now) { t_target = t_target_now; lambda_temp = lambda_temp_now; }

 protected:
  double t_start, t_stop, t_period, t_target, lambda_temp = 0.0;
  int seed;
  bool created = false;
  // fix_modify temp (UCG/fix_ucgld_langevin.cpp:380-398) and the bias flag init() derives from it (:162-165)
  char *id_temp = nullptr;
  class Compute *temperature = nullptr;
  int tbiasflag = 0;
  int nlevels_respa = 1;
};

class FixUCGStateGPU : public FixUCGGPUBase {
 public:
  FixUCGStateGPU(class LAMMPS *, int, char **);
  int setmask() override;
  void init() override;
  void setup(int) override;
  void post_force(int) override;

 protected:
  int ld_flag = 0, mc_flag = 0, mc_seed = 0;
  double mc_rate = 0.01;
  bool created = false;
};

// fix cluster_switch needs the RESIDENT lists (the device-built full list and ghosts): it runs inside the library's step
// loop, i.e. under `run_style verlet/ucg/gpu`, where ucg_md_run does its pre_exchange work at the forced re-neighbour steps
// and -- on several ranks -- the MPI_Allreduce steps of UCG/fix_cluster_switch.cpp:114-120, 157-158, 664, 750 through the
// attached communicator.  Under the stock run_style the reference's own CPU fix cluster_switch keeps working unchanged on
// LAMMPS' arrays (it only changes atom->type, which the pair style uploads at the next re-neighbour step).
class FixClusterSwitchGPU : public FixUCGGPUBase {
 public:
  FixClusterSwitchGPU(class LAMMPS *, int, char **);
  int setmask() override;
  void init() override;
  double compute_vector(int) override;
  // called by run_style verlet/ucg/gpu once the atoms and their molecule ids are on the device
  void create_on_device(ucg_ctx *);

 protected:
  bool created = false;
  int mol_seed, mol_offset, seed, switchFreq;
  double cutoff;
  std::string rateFile, contactFile;
};

}    // namespace LAMMPS_NS
#endif
#endif
