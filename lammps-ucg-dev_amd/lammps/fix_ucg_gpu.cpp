// USER-UCG/GPU fixes: LAMMPS-side binding of libucg_hip.so -- see fix_ucg_gpu.h.
//
// Hook by hook the classes below do what the reference's do (file:line in each method); the
// arithmetic itself runs in the HIP kernels of csrc/ucg_fix.hip through the C ABI.
#include "fix_ucg_gpu.h"

#include "atom.h"
#include "atom_vec_ucg_gpu.h"
#include "comm.h"
#include "compute.h"
#include "error.h"
#include "force.h"
#include "group.h"
#include "modify.h"
#include "neighbor.h"
#include "output.h"
#include "pair.h"
#include "respa.h"
#include "update.h"
#include "utils.h"

#include "ucg_hip.h"

#include <cstring>
#include <string>
#include <vector>

using namespace LAMMPS_NS;
using namespace FixConst;

/* ------------------------------------------------------------------ shared plumbing */

FixUCGGPUBase::FixUCGGPUBase(LAMMPS *lmp, int narg, char **arg) : Fix(lmp, narg, arg)
{
  AtomVecUCG::get(lmp);    // "requires ucg atom style" (UCG/fix_ucgstate.cpp:41-43)
  dynamic_group_allow = 1;
}

void FixUCGGPUBase::init()
{
  // one device context per rank: the pair style's (created in its constructor)
  int dim = 0;
  ctx = force->pair ? (ucg_ctx *) force->pair->extract("ucg_ctx", dim) : nullptr;
  if (!ctx) error->all(FLERR, "USER-UCG/GPU fixes need one of the pair styles table_ucgld / table_ucg_bethe / table_ucg_bethe_density");
  check(ucg_ctx_set_units(ctx, force->boltz, force->ftm2v, force->mvv2e, update->dt, force->special_lj));
  const int *r = (const int *) force->pair->extract("ucg_resident", dim);
  resident = r ? *r : 0;
  const int *d = (const int *) force->pair->extract("ucg_driver", dim);
  driver = d ? *d : 0;
}

void FixUCGGPUBase::check(int rc)
{
  if (rc != UCG_OK) error->one(FLERR, ucg_last_error(ctx));
}

void FixUCGGPUBase::to_device(int fields)
{
  if (resident) return;    // the device copy is the authoritative one (host edits are announced with ucg_host_modified)
  auto avec = AtomVecUCG::get(lmp);
  check(ucg_atoms_upload_owned(ctx, (fields & X) ? &atom->x[0][0] : nullptr, (fields & V) ? &atom->v[0][0] : nullptr,
                               (fields & F) ? &atom->f[0][0] : nullptr, (fields & STATE) ? avec->ucgstate : nullptr,
                               (fields & NSTATES) ? avec->num_ucgstates : nullptr, (fields & L) ? avec->ucgl : nullptr,
                               (fields & VL) ? avec->ucgvl : nullptr, (fields & P) ? avec->ucgp : nullptr,
                               (fields & LF) ? avec->ucgforce : nullptr,
                               (fields & SCORES) ? &avec->ucgsoftmaxscores[0][0] : nullptr));
}

void FixUCGGPUBase::from_device(int fields)
{
  if (resident) return;    // LAMMPS' arrays catch up in pre_exchange() / on output steps
  auto avec = AtomVecUCG::get(lmp);
  check(ucg_atoms_download(ctx, 0, (fields & X) ? &atom->x[0][0] : nullptr, (fields & V) ? &atom->v[0][0] : nullptr,
                           nullptr, nullptr, nullptr, (fields & STATE) ? avec->ucgstate : nullptr, nullptr,
                           (fields & L) ? avec->ucgl : nullptr, (fields & VL) ? avec->ucgvl : nullptr, nullptr,
                           (fields & P) ? avec->ucgp : nullptr, (fields & LF) ? avec->ucgforce : nullptr, nullptr));
}

/* ------------------------------------------------------------------ fix nve/ucgld */

FixNVEUCGLDGPU::FixNVEUCGLDGPU(LAMMPS *lmp, int narg, char **arg) : FixUCGGPUBase(lmp, narg, arg)
{
  // UCG/fix_nve_ucgld.cpp:13-20
  if (utils::strmatch(style, "^nve/ucgld$") && narg > 3)
    error->all(FLERR, 3, "Unsupported additional arguments for fix {}", style);
  time_integrate = 1;
}

FixNVEUCGLDGPU::~FixNVEUCGLDGPU()
{
  restore_neigh_modify();
}

int FixNVEUCGLDGPU::setmask()
{
  // UCG/fix_nve_ucgld.cpp:27-34, plus the hooks of the resident mode (no-ops otherwise)
  return INITIAL_INTEGRATE | FINAL_INTEGRATE | INITIAL_INTEGRATE_RESPA | FINAL_INTEGRATE_RESPA | PRE_EXCHANGE | END_OF_STEP | POST_RUN;
}

void FixNVEUCGLDGPU::restore_neigh_modify()
{
  if (!neigh_taken) return;
  neighbor->delay = saved_delay;
  neighbor->every = saved_every;
  neigh_taken = false;
}

void FixNVEUCGLDGPU::post_run()
{
  restore_neigh_modify();
}

void FixNVEUCGLDGPU::init()
{
  FixUCGGPUBase::init();
  if (utils::strmatch(update->integrate_style, "^respa")) step_respa = (dynamic_cast<Respa *>(update->integrate))->step;
  // atom style ucg has per-type masses (mass_type = PER_TYPE, UCG/atom_vec_ucg.cpp:36), so the reference's rmass branch
  // (UCG/fix_nve_ucgld.cpp:64-78, 124-138) is reachable only through atom_style hybrid with a per-atom-mass style
  if (atom->rmass) error->all(FLERR, "USER-UCG/GPU integrators use the per-type masses of atom style ucg");
  restore_neigh_modify();    // (a run that ended in an error never reached post_run)
  if (resident) {
    // Neighbor::decide() would read x on the host every `every` steps: the distance check runs on the device instead
    // (ucg_decide_local, with the neigh_modify settings the pair style saved in init_style(), which ran before this) and a
    // positive outcome is handed to LAMMPS through force_reneighbor / next_reneighbor; LAMMPS' own criterion is switched
    // off by an unreachable delay FOR THIS RUN ONLY: post_run() puts the user's values back (left in
    // place they reached the pair style's next init_style() as "the user's delay" and the device never re-neighboured
    // again in a second `run`)
    force_reneighbor = 1;
    next_reneighbor = -1;
    saved_delay = neighbor->delay;
    saved_every = neighbor->every;
    neigh_taken = true;
    neighbor->delay = UCG_GPU_DELAY_SENTINEL;
    neighbor->every = 1;
  } else {
    force_reneighbor = 0;
  }
}

void FixNVEUCGLDGPU::pre_exchange()
{
  // a re-neighbouring step: pbc / exchange / borders work on LAMMPS' arrays
  if (resident) check(ucg_host_sync(ctx, UCG_F_X | UCG_F_V | UCG_F_STATE | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_UCGP));
}

void FixNVEUCGLDGPU::end_of_step()
{
  // thermo / dump / restart output of this step reads LAMMPS' arrays
  if (resident && update->ntimestep == output->next) check(ucg_host_sync(ctx, UCG_F_ALL));
}

void FixNVEUCGLDGPU::set_step(double dt)
{
  // dtv = dt, dtf = 0.5 * dt * ftm2v are formed inside the library from the context's dt
  check(ucg_ctx_set_units(ctx, force->boltz, force->ftm2v, force->mvv2e, dt, force->special_lj));
}

void FixNVEUCGLDGPU::reset_dt()
{
  set_step(update->dt);
}

// rRESPA (UCG/fix_nve_ucgld.cpp:155-173): the level's step; the innermost level moves x and lambda, the others only kick
void FixNVEUCGLDGPU::initial_integrate_respa(int vflag, int ilevel, int /*iloop*/)
{
  set_step(step_respa[ilevel]);
  if (ilevel == 0) initial_integrate(vflag);
  else final_integrate();
}

void FixNVEUCGLDGPU::final_integrate_respa(int ilevel, int /*iloop*/)
{
  set_step(step_respa[ilevel]);
  final_integrate();
}

void FixNVEUCGLDGPU::initial_integrate(int)
{
  // UCG/fix_nve_ucgld.cpp:44-101 (wall/hard: UCG/fix_nve_ucgld_wall_hard.cpp:61-136)
  to_device(X | V | F | L | VL | LF | (wall ? STATE : 0));
  check(wall ? ucg_fix_nve_wall_hard_initial(ctx, groupbit) : ucg_fix_nve_initial(ctx, groupbit));
  from_device(X | V | L | VL | (wall ? STATE : 0));
  if (resident) {
    int due = 0, flag = 0;
    check(ucg_md_set_timestep(ctx, update->ntimestep));
    check(ucg_decide_local(ctx, &due, &flag));
    if (due && flag) next_reneighbor = update->ntimestep;    // Neighbor::decide() returns 1 for this step
  }
}

void FixNVEUCGLDGPU::final_integrate()
{
  // UCG/fix_nve_ucgld.cpp:104-153 (wall/hard: UCG/fix_nve_ucgld_wall_hard.cpp:139-199)
  to_device(V | F | L | VL | LF);
  check(wall ? ucg_fix_nve_wall_hard_final(ctx, groupbit) : ucg_fix_nve_final(ctx, groupbit));
  from_device(V | VL | (wall ? L : 0));
}

/* ------------------------------------------------------------------ fix nve/ucgld/wall/hard */

FixNVEUCGLDWallHardGPU::FixNVEUCGLDWallHardGPU(LAMMPS *lmp, int narg, char **arg) : FixNVEUCGLDGPU(lmp, narg, arg)
{
  // UCG/fix_nve_ucgld_wall_hard.cpp:12-34
  if (narg > 5) error->all(FLERR, 3, "Unsupported additional arguments for fix {}", style);
  wall = true;
  int iarg = 3;
  while (iarg < narg) {
    if (utils::strmatch(arg[iarg], "bias_potential")) {
      bias_potential_flag = 1;
      iarg++;
      if (iarg < narg) barrier = utils::numeric(FLERR, arg[iarg], false, lmp);
      iarg++;
    } else
      error->all(FLERR, "Unknown argument for fix {}", style);
  }
}

int FixNVEUCGLDWallHardGPU::setmask()
{
  int mask = INITIAL_INTEGRATE | FINAL_INTEGRATE;
  if (bias_potential_flag) mask |= POST_FORCE;    // :41-52
  return mask;
}

void FixNVEUCGLDWallHardGPU::init()
{
  FixNVEUCGLDGPU::init();
  check(ucg_fix_nve_wall_hard_set(ctx, bias_potential_flag, barrier));
}

void FixNVEUCGLDWallHardGPU::post_force(int)
{
  // :223-241
  to_device(L | LF | F);
  check(ucg_fix_nve_wall_hard_post_force(ctx, groupbit));
  from_device(LF);
}

/* ------------------------------------------------------------------ fix ucgld/langevin */

FixUCGLDLangevinGPU::FixUCGLDLangevinGPU(LAMMPS *lmp, int narg, char **arg) : FixUCGGPUBase(lmp, narg, arg)
{
  // UCG/fix_ucgld_langevin.cpp:54-84
  if (narg < 7) error->all(FLERR, "Illegal fix langevin command");
  scalar_flag = 1;
  global_freq = 1;
  extscalar = 1;
  nevery = 1;
  if (utils::strmatch(arg[3], "^v_"))
    error->all(FLERR, "lambda dynamic variable is not supported with variable temperature");
  t_start = utils::numeric(FLERR, arg[3], false, lmp);
  t_target = t_start;
  t_stop = utils::numeric(FLERR, arg[4], false, lmp);
  t_period = utils::numeric(FLERR, arg[5], false, lmp);
  seed = utils::inumeric(FLERR, arg[6], false, lmp);
  if (t_period <= 0.0) error->all(FLERR, "Fix langevin period must be > 0.0");
  if (seed <= 0) error->all(FLERR, "Illegal fix langevin command");
}

FixUCGLDLangevinGPU::~FixUCGLDLangevinGPU()
{
  delete[] id_temp;
}

int FixUCGLDLangevinGPU::setmask()
{
  return POST_FORCE | POST_FORCE_RESPA | END_OF_STEP;    // :137-144
}

void FixUCGLDLangevinGPU::init()
{
  FixUCGGPUBase::init();
  // :151-160, :173-176: the compute of fix_modify temp and whether it removes a velocity bias
  tbiasflag = 0;
  if (id_temp) {
    temperature = modify->get_compute_by_id(id_temp);
    if (!temperature) error->all(FLERR, "Temperature compute ID {} for fix {} does not exist", id_temp, style);
    if (temperature->tempflag == 0) error->all(FLERR, "Compute ID {} for fix {} does not compute temperature", id_temp, style);
    if (temperature->tempbias) tbiasflag = 1;
  }
  if (utils::strmatch(update->integrate_style, "^respa")) nlevels_respa = (dynamic_cast<Respa *>(update->integrate))->nlevels;
  if (!created) {
    // RanMars(seed + comm->me), :85
    check(ucg_fix_langevin_create(ctx, t_start, t_stop, t_period, seed, comm->me));
    created = true;
  }
  // :149-183: the prefactors read atom->ucgml[TYPE INDEX] -- same here
  auto avec = AtomVecUCG::get(lmp);
  std::vector<double> ml((size_t) atom->ntypes + 1, 0.0);
  for (int i = 1; i <= atom->ntypes; i++) ml[(size_t) i] = (i < atom->nlocal) ? avec->ucgml[i] : (atom->nlocal ? avec->ucgml[0] : 1.0);
  check(ucg_fix_langevin_init_from_ucgml(ctx, atom->ntypes, ml.data()));
  check(ucg_fix_langevin_set_bias(ctx, tbiasflag));    // post_force_templated<1> when BIAS (:203-210)
}

void FixUCGLDLangevinGPU::setup(int vflag)
{
  // :187-197
  if (utils::strmatch(update->integrate_style, "^verlet")) post_force(vflag);
  else {
    auto respa = dynamic_cast<Respa *>(update->integrate);
    respa->copy_flevel_f(nlevels_respa - 1);
    post_force_respa(vflag, nlevels_respa - 1, 0);
    respa->copy_f_flevel(nlevels_respa - 1);
  }
}

void FixUCGLDLangevinGPU::post_force_respa(int vflag, int ilevel, int /*iloop*/)
{
  if (ilevel == nlevels_respa - 1) post_force(vflag);    // :216-219
}

void FixUCGLDLangevinGPU::reset_target(double t_new)
{
  t_target = t_start = t_stop = t_new;    // :358-361
  if (ctx && created) check(ucg_fix_langevin_reset_target(ctx, t_new));
}

void FixUCGLDLangevinGPU::reset_dt()
{
  // :366-376 AS SHIPPED: gfactor2 from atom->mass[type] (init() used atom->ucgml[type index]), gfactor1 untouched.  Before
  // the first init() there is nothing to reset: init() forms both prefactors from update->dt.
  if (!ctx || !created || !atom->mass) return;
  check(ucg_ctx_set_units(ctx, force->boltz, force->ftm2v, force->mvv2e, update->dt, force->special_lj));
  check(ucg_fix_langevin_reset_dt(ctx, atom->ntypes, atom->mass));
}

int FixUCGLDLangevinGPU::modify_param(int narg, char **arg)
{
  // :380-398
  if (strcmp(arg[0], "temp") == 0) {
    if (narg < 2) utils::missing_cmd_args(FLERR, "fix_modify", error);
    delete[] id_temp;
    id_temp = utils::strdup(arg[1]);
    temperature = modify->get_compute_by_id(id_temp);
    if (!temperature) error->all(FLERR, "Could not find fix_modify temperature compute ID: {}", id_temp);
    if (temperature->tempflag == 0) error->all(FLERR, "Fix_modify temperature compute {} does not compute temperature", id_temp);
    if (temperature->igroup != igroup && comm->me == 0)
      error->warning(FLERR, "Group for fix_modify temp != fix group: {} vs {}", group->names[igroup], group->names[temperature->igroup]);
    return 2;
  }
  return 0;
}

void FixUCGLDLangevinGPU::post_force(int)
{
  // :226-297 (compute_target :318-353 inside the library: needs the run's begin/end steps)
  if (tbiasflag) {
    // post_force_templated<1> evaluates the bias compute first (:268); it reads LAMMPS' velocities
    if (resident) check(ucg_host_sync(ctx, UCG_F_V));
    temperature->compute_scalar();
  }
  to_device(VL | LF | F);
  check(ucg_fix_langevin_post_force(ctx, groupbit, update->ntimestep, update->beginstep, update->endstep));
  t_target = ucg_fix_langevin_t_target(ctx);
  from_device(LF);
}

void FixUCGLDLangevinGPU::end_of_step()
{
  // :303-312: kinetic temperature of lambda
  to_device(VL | V);
  check(ucg_fix_langevin_end_of_step(ctx, groupbit, &lambda_temp));
}

double FixUCGLDLangevinGPU::compute_scalar()
{
  return lambda_temp;    // :403-406
}

void *FixUCGLDLangevinGPU::extract(const char *str, int &dim)
{
  // :412-417: polled by the pair styles and by fix ucgstate
  dim = 0;
  if (strcmp(str, "t_target") == 0) return &t_target;
  return nullptr;
}

/* ------------------------------------------------------------------ fix ucgstate */

FixUCGStateGPU::FixUCGStateGPU(LAMMPS *lmp, int narg, char **arg) : FixUCGGPUBase(lmp, narg, arg)
{
  // UCG/fix_ucgstate.cpp:24-67
  if (narg > 6) error->all(FLERR, 3, "Too many arguments for fix {}", style);
  if (narg > 3) {
    if (utils::strmatch(arg[3], "ld")) ld_flag = 1;
    else if (utils::strmatch(arg[3], "mc")) {
      mc_flag = 1;
      if (narg == 4) error->all(FLERR, 1, "fix ucgstate mc requires seed and rate information");
      if (narg == 5) error->all(FLERR, 1, "fix ucgstate mc requires rate information");
      mc_seed = utils::inumeric(FLERR, arg[4], false, lmp);
      mc_rate = utils::numeric(FLERR, arg[5], false, lmp);
    } else
      error->all(FLERR, 1, "Unknown argument for fix {}: {}", style, arg[3]);
  }
  time_integrate = 0;
}

int FixUCGStateGPU::setmask()
{
  return POST_FORCE | MIN_POST_FORCE;    // :75-81 without the rRESPA hook
}

void FixUCGStateGPU::init()
{
  FixUCGGPUBase::init();
  if (!created) {
    check(ucg_fix_ucgstate_create(ctx, ld_flag, mc_flag, mc_seed, mc_rate, comm->me));    // RanMars(mc_seed + me), :57
    created = true;
  }
}

void FixUCGStateGPU::setup(int vflag)
{
  // :142-171: a thermostat exporting t_target must be defined BEFORE this fix
  double *pT = nullptr;
  int pdim;
  for (int ifix = 0; ifix < modify->nfix; ifix++) {
    pT = (double *) modify->fix[ifix]->extract("t_target", pdim);
    if (pT) break;
  }
  if (pT == nullptr)
    error->all(FLERR, "FixUCGState requires a thermostat fix BEFORE ITSELF to set the target temperature T.");
  post_force(vflag);
}

void FixUCGStateGPU::post_force(int)
{
  // :88-132
  to_device(SCORES | NSTATES | STATE | L);
  check(ucg_fix_ucgstate_post_force(ctx));
  from_device(P | (ld_flag ? 0 : (STATE | L)));
}

/* ------------------------------------------------------------------ fix cluster_switch */

FixClusterSwitchGPU::FixClusterSwitchGPU(LAMMPS *lmp, int narg, char **arg) : FixUCGGPUBase(lmp, narg, arg)
{
  // UCG/fix_cluster_switch.cpp:37-60: arg[3] molID_seed, [4] mol_offset, [5] cutoff, [6] seed,
  // [7] "rateFreq" [8] N, [9] "rateFile" [10] file, [11] "contactFile" [12] file
  if (narg < 13) error->all(FLERR, "Illegal cluster_switch command");
  mol_seed = utils::inumeric(FLERR, arg[3], false, lmp);
  mol_offset = utils::inumeric(FLERR, arg[4], false, lmp);
  cutoff = utils::numeric(FLERR, arg[5], false, lmp);
  seed = utils::inumeric(FLERR, arg[6], false, lmp);
  switchFreq = utils::inumeric(FLERR, arg[8], false, lmp);
  rateFile = arg[10];
  contactFile = arg[12];
  if (atom->molecule_flag == 0) error->all(FLERR, "fix cluster_switch requires that atoms have molecule attributes");
  vector_flag = 1;
  size_vector = 7;
  global_freq = 1;
  extvector = 0;
  force_reneighbor = 1;
}

int FixClusterSwitchGPU::setmask()
{
  return PRE_EXCHANGE;    // :347-352; the work itself runs inside the resident loop (ucg_md_run)
}

void FixClusterSwitchGPU::init()
{
  FixUCGGPUBase::init();
  if (!driver)
    error->all(FLERR, "USER-UCG/GPU fix cluster_switch works on the device-built lists of the resident step loop: use "
                      "`run_style verlet/ucg/gpu` (or the reference's CPU fix cluster_switch with the stock run_style)");
  // the device objects are made by the run style's setup(), once the atoms and their molecule ids are uploaded
}

void FixClusterSwitchGPU::create_on_device(ucg_ctx *c)
{
  // the constructor's survey (UCG/fix_cluster_switch.cpp:62-160) needs the atoms: once, like the reference's constructor;
  // on several ranks the library reduces it over the attached communicator at the next ucg_md_setup
  if (created) return;
  ctx = c;
  check(ucg_fix_cluster_switch_create(ctx, groupbit, mol_seed, mol_offset, cutoff, seed, switchFreq, rateFile.c_str(),
                                      contactFile.c_str()));
  created = true;
}

double FixClusterSwitchGPU::compute_vector(int n)
{
  // :887-897
  double v[7];
  check(ucg_fix_cluster_switch_vector(ctx, v));
  return (n >= 0 && n < 7) ? v[n] : 0.0;
}
