// run_style verlet/ucg/gpu: the step loop of an all-USER-UCG/GPU deck inside libucg_hip.so -- see verlet_ucg_gpu.h.
//
// What it stands in for, call by call (upstream Verlet, restated in SURVEY.md section 3.1):
//   Verlet::setup()    pbc / exchange / borders / neighbour build / force_clear / Pair::compute / Modify::setup
//                      -> upload_atoms() + ucg_md_setup()                (one rank: csrc/ucg_neigh.hip; several: csrc/ucg_comm.hip)
//   Verlet::run(n)     per step initial_integrate / decide / [re-neighbour | forward_comm] / compute / post_force /
//                      final_integrate / end_of_step
//                      -> ucg_md_run_until(chunk) between two output steps, download_atoms() + Output::write on them
//   Verlet::cleanup()  -> download_atoms(): LAMMPS' arrays are authoritative again between two `run` commands
#include "verlet_ucg_gpu.h"

#include "atom.h"
#include "atom_vec_ucg_gpu.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "fix_ucg_gpu.h"
#include "force.h"
#include "modify.h"
#include "neighbor.h"
#include "output.h"
#include "pair_table_ucg_gpu.h"
#include "timer.h"
#include "update.h"
#include "utils.h"

#include "ucg_hip.h"

#include <climits>
#include <cstring>
#include <vector>

using namespace LAMMPS_NS;

VerletUCGGPU::VerletUCGGPU(LAMMPS *lmp, int narg, char **arg) : Integrate(lmp, narg, arg)
{
  // run_style verlet/ucg/gpu [comm rccl|mpi]
  int iarg = 0;
  while (iarg < narg) {
    if (strcmp(arg[iarg], "comm") == 0 && iarg + 1 < narg) {
      if (strcmp(arg[iarg + 1], "rccl") == 0) use_rccl = 1;
      else if (strcmp(arg[iarg + 1], "mpi") == 0) use_rccl = 0;
      else error->all(FLERR, "run_style verlet/ucg/gpu comm: expected rccl or mpi, not {}", arg[iarg + 1]);
      iarg += 2;
    } else
      error->all(FLERR, "Unknown run_style verlet/ucg/gpu argument: {}", arg[iarg]);
  }
}

VerletUCGGPU::~VerletUCGGPU()
{
  // the context is the pair style's; detach what this class attached to it (the pair style may outlive the run style)
  if (attached && ctx) ucg_comm_detach(ctx);
  if (have_gridworld) MPI_Comm_free(&gridworld);
}

void VerletUCGGPU::check(int rc)
{
  // every rank returns from the library's collectives together (include/ucg_hip.h, "communicator of a decomposed run"),
  // with an error code on ALL of them when any failed: error->one on each is then a clean stop
  if (rc != UCG_OK) error->one(FLERR, ucg_last_error(ctx));
}

void VerletUCGGPU::find_styles()
{
  pair = dynamic_cast<PairTableUCGGPU *>(force->pair);
  if (!pair)
    error->all(FLERR, "run_style verlet/ucg/gpu needs pair_style table_ucgld, table_ucg_bethe or table_ucg_bethe_density of USER-UCG/GPU");
  ctx = pair->device_context();
  gpair = pair->device_pair();
  use_nve = use_lang = use_ucgst = 0;
  fix_lang = nullptr;
  fix_cs = nullptr;
  for (int i = 0; i < modify->nfix; i++) {
    Fix *f = modify->fix[i];
    if (auto w = dynamic_cast<FixNVEUCGLDGPU *>(f)) {
      if (use_nve) error->all(FLERR, "run_style verlet/ucg/gpu: more than one time-integration fix");
      use_nve = w->is_wall() ? 2 : 1;
    } else if (auto l = dynamic_cast<FixUCGLDLangevinGPU *>(f)) {
      if (use_lang) error->all(FLERR, "run_style verlet/ucg/gpu: more than one fix ucgld/langevin");
      if (use_ucgst) error->all(FLERR, "FixUCGState requires a thermostat fix BEFORE ITSELF to set the target temperature T.");
      use_lang = 1;
      fix_lang = l;
    } else if (dynamic_cast<FixUCGStateGPU *>(f)) {
      if (use_ucgst) error->all(FLERR, "run_style verlet/ucg/gpu: more than one fix ucgstate");
      use_ucgst = 1;
    } else if (auto c = dynamic_cast<FixClusterSwitchGPU *>(f)) {
      fix_cs = c;
    } else {
      // any other fix may read or write LAMMPS' arrays at a hook this run style never reaches
      error->all(FLERR, "run_style verlet/ucg/gpu: fix {} (style {}) is not of USER-UCG/GPU; use run_style verlet", f->id, f->style);
    }
    // one group for all of them: the library's loop applies a single group bit
    if (dynamic_cast<FixNVEUCGLDGPU *>(f) || dynamic_cast<FixUCGLDLangevinGPU *>(f))
      if (f->igroup != 0) error->all(FLERR, "run_style verlet/ucg/gpu: fix {} must be applied to group all", f->id);
  }
}

void VerletUCGGPU::init()
{
  Integrate::init();
  find_styles();
  if (domain->triclinic) error->all(FLERR, "run_style verlet/ucg/gpu needs an orthogonal box");
  if (!domain->xperiodic || !domain->yperiodic || !domain->zperiodic) error->all(FLERR, "run_style verlet/ucg/gpu needs a periodic box");
  if (comm->layout != Comm::LAYOUT_UNIFORM)
    error->all(FLERR, "run_style verlet/ucg/gpu needs the uniform processor grid (no fix balance / comm_style tiled)");
  if (atom->rmass) error->all(FLERR, "USER-UCG/GPU integrators use the per-type masses of atom style ucg");
  if (atom->q_flag) {
    int nonzero = 0, any = 0;
    for (int i = 0; i < atom->nlocal; i++) nonzero |= (atom->q[i] != 0.0);
    MPI_Allreduce(&nonzero, &any, 1, MPI_INT, MPI_MAX, world);
    if (any) error->all(FLERR, "run_style verlet/ucg/gpu: charges do not travel with the device-side migration; the UCG styles do not use them -- set them to zero");
  }
  // the library's rank numbering over comm->procgrid (x fastest), whatever `processors ... map` made of MPI's
  me_grid = comm->myloc[0] + comm->procgrid[0] * (comm->myloc[1] + comm->procgrid[1] * comm->myloc[2]);
}

/* ---------------------------------------------------------------- communicator */

int VerletUCGGPU::cb_alltoallv(void *user, const void *send, const long long *sb, void *recv, const long long *rb, void *)
{
  auto self = static_cast<VerletUCGGPU *>(user);
  const int w = self->comm->nprocs;
  std::vector<int> sc((size_t) w), sd((size_t) w), rc((size_t) w), rd((size_t) w);
  long long so = 0, ro = 0;
  for (int r = 0; r < w; r++) {
    if (sb[r] > INT_MAX || rb[r] > INT_MAX || so > INT_MAX || ro > INT_MAX) return 1;    // (2 GB per peer: 40 M halo records)
    sc[(size_t) r] = (int) sb[r];
    sd[(size_t) r] = (int) so;
    rc[(size_t) r] = (int) rb[r];
    rd[(size_t) r] = (int) ro;
    so += sb[r];
    ro += rb[r];
  }
  return MPI_Alltoallv(send, sc.data(), sd.data(), MPI_BYTE, recv, rc.data(), rd.data(), MPI_BYTE, self->gridworld) == MPI_SUCCESS ? 0 : 1;
}

int VerletUCGGPU::cb_alltoall_ll(void *user, const long long *send, long long *recv)
{
  auto self = static_cast<VerletUCGGPU *>(user);
  return MPI_Alltoall(send, 1, MPI_LONG_LONG, recv, 1, MPI_LONG_LONG, self->gridworld) == MPI_SUCCESS ? 0 : 1;
}

int VerletUCGGPU::cb_allreduce_ll(void *user, long long *buf, int n, int op)
{
  auto self = static_cast<VerletUCGGPU *>(user);
  return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_LONG_LONG, op == 0 ? MPI_SUM : (op == 1 ? MPI_MAX : MPI_MIN), self->gridworld) == MPI_SUCCESS ? 0 : 1;
}

int VerletUCGGPU::cb_allreduce_f64(void *user, double *buf, int n, int op)
{
  auto self = static_cast<VerletUCGGPU *>(user);
  return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, op == 0 ? MPI_SUM : (op == 1 ? MPI_MAX : MPI_MIN), self->gridworld) == MPI_SUCCESS ? 0 : 1;
}

void VerletUCGGPU::attach_communicator()
{
  if (attached || comm->nprocs == 1) return;
  if (!have_gridworld) {
    MPI_Comm_split(world, 0, me_grid, &gridworld);    // rank in gridworld == me_grid
    have_gridworld = true;
  }
  if (use_rccl) {
    // rank 0 of the grid makes the id, MPI carries its 128 bytes, every rank joins (ncclCommInitRank inside)
    ucg_rccl_id id;
    memset(&id, 0, sizeof(id));
    int ok = 1;
    if (me_grid == 0) ok = (ucg_comm_rccl_unique_id(&id) == UCG_OK) ? 1 : 0;
    MPI_Bcast(&ok, 1, MPI_INT, 0, gridworld);
    if (!ok) error->all(FLERR, "run_style verlet/ucg/gpu: librccl could not be loaded (use `run_style verlet/ucg/gpu comm mpi`)");
    MPI_Bcast(&id, (int) sizeof(id), MPI_BYTE, 0, gridworld);
    int rc = ucg_comm_attach_rccl(ctx, &id, me_grid, comm->nprocs), worst = 0;
    MPI_Allreduce(&rc, &worst, 1, MPI_INT, MPI_MAX, gridworld);
    if (worst != UCG_OK)
      error->all(FLERR, "run_style verlet/ucg/gpu: RCCL did not attach on every rank ({}); one GPU per rank is required -- or use `comm mpi`",
                 rc != UCG_OK ? ucg_last_error(ctx) : "another rank failed");
  } else {
    ucg_comm_ops ops;
    ops.user = this;
    ops.rank = me_grid;
    ops.world = comm->nprocs;
    ops.alltoallv = cb_alltoallv;
    ops.alltoall_ll = cb_alltoall_ll;
    ops.allreduce_ll = cb_allreduce_ll;
    ops.allreduce_f64 = cb_allreduce_f64;
    check(ucg_comm_attach_host(ctx, &ops));
  }
  attached = true;
}

/* ---------------------------------------------------------------- data movement at the ends of a run */

void VerletUCGGPU::upload_atoms()
{
  auto avec = AtomVecUCG::get(lmp);
  const int nlocal = atom->nlocal;
  // owned atoms only: the library finds the owner rank of every bead itself (its exchange sends a bead from wherever it is
  // to the brick that holds its wrapped position) and builds the ghosts on the device
  check(ucg_atoms_upload(ctx, nlocal, 0, atom->ntypes, nlocal ? &atom->x[0][0] : nullptr, nlocal ? &atom->v[0][0] : nullptr, atom->type,
                         atom->tag, atom->mask, avec->ucgstate, avec->ucgl, avec->ucgvl, avec->ucgml, avec->ucgp, atom->mass));
  if (atom->molecule_flag) {
    std::vector<int> mol((size_t) nlocal + 1);
    for (int i = 0; i < nlocal; i++) mol[(size_t) i] = (int) atom->molecule[i];
    check(ucg_atoms_upload_molecule(ctx, mol.data()));
  }
  // neigh_modify every / delay / check and the skin as the input deck gave them: Neighbor::decide() runs on the device
  check(ucg_ctx_set_units(ctx, force->boltz, force->ftm2v, force->mvv2e, update->dt, force->special_lj));
  check(ucg_domain_set(ctx, domain->boxlo, domain->boxhi, ucg_pair_cutforce(gpair), neighbor->skin, neighbor->every, neighbor->delay,
                       neighbor->dist_check));
  if (comm->nprocs > 1) check(ucg_decomp_set(ctx, comm->procgrid, me_grid));
}

void VerletUCGGPU::download_atoms(bool with_energy)
{
  auto avec = AtomVecUCG::get(lmp);
  int nlocal = 0, nghost = 0;
  check(ucg_atoms_counts(ctx, &nlocal, &nghost));
  if (nlocal > atom->nmax) atom->avec->grow(nlocal);    // beads migrated in (grow_pointers() refreshes avec's pointers)
  atom->nlocal = nlocal;
  atom->nghost = 0;    // LAMMPS holds no ghosts under this run style
  check(ucg_atoms_download(ctx, 0, nlocal ? &atom->x[0][0] : nullptr, nlocal ? &atom->v[0][0] : nullptr, nlocal ? &atom->f[0][0] : nullptr,
                           atom->type, atom->tag, avec->ucgstate, avec->num_ucgstates, avec->ucgl, avec->ucgvl, avec->ucgml, avec->ucgp,
                           avec->ucgforce, nlocal ? &avec->ucgsoftmaxscores[0][0] : nullptr));
  check(ucg_atoms_download_mask(ctx, atom->mask));
  if (atom->molecule_flag) {
    std::vector<int> mol((size_t) nlocal + 1);
    check(ucg_atoms_download_molecule(ctx, mol.data()));
    for (int i = 0; i < nlocal; i++) atom->molecule[i] = mol[(size_t) i];
  }
  // positions come back wrapped into the box; image counts are not tracked on the device
  const imageint img0 = ((imageint) IMGMAX << IMG2BITS) | ((imageint) IMGMAX << IMGBITS) | IMGMAX;
  for (int i = 0; i < nlocal; i++) atom->image[i] = img0;
  if (atom->q_flag)
    for (int i = 0; i < nlocal; i++) atom->q[i] = 0.0;
  if (atom->map_style != Atom::MAP_NONE) {
    atom->map_init();
    atom->map_set();
  }
  if (with_energy) {
    // totals over all ranks are on every rank (ucg_md_thermo); compute pe / compute pressure sum Pair::eng_vdwl and
    // Pair::virial over the ranks, so rank 0 reports the totals and the others zero
    double th[9];
    check(ucg_md_thermo(ctx, th));
    const double share = (comm->me == 0) ? 1.0 : 0.0;
    pair->eng_vdwl = share * th[0];
    for (int k = 0; k < 6; k++) pair->virial[k] = share * th[1 + k];
    if (fix_lang) fix_lang->set_from_driver(ucg_fix_langevin_t_target(ctx), th[7]);
  }
}

/* ---------------------------------------------------------------- Integrate interface */

void VerletUCGGPU::device_setup()
{
  // Verlet::setup(): ev_set() tells compute pe / compute pressure that energy and virial are tallied on this step
  // (update->eflag_global / vflag_global; without it thermo stops with "Energy was not tallied on needed timestep")
  ev_set(update->ntimestep);
  upload_atoms();
  attach_communicator();
  if (fix_cs) fix_cs->create_on_device(ctx);
  check(ucg_md_attach(ctx, gpair, use_nve, use_lang, use_ucgst));
  check(ucg_md_set_timestep(ctx, update->ntimestep));
  check(ucg_md_setup(ctx, update->laststep - update->ntimestep));
  // `run N start S stop E`: the ramp of the thermostat's target runs over [beginstep, endstep]
  check(ucg_md_set_window(ctx, update->beginstep, update->endstep));
  download_atoms(true);
}

void VerletUCGGPU::setup(int flag)
{
  if (comm->me == 0 && screen) fputs("Setting up verlet/ucg/gpu run ...\n", screen);
  update->setupflag = 1;
  device_setup();
  output->setup(flag);
  update->setupflag = 0;
}

void VerletUCGGPU::setup_minimal(int /*flag*/)
{
  // Verlet::setup_minimal: forces (and, with flag, the lists) without output.  LAMMPS' arrays are the authoritative copy
  // outside a run, so the device state is always made afresh from them -- lists included
  update->setupflag = 1;
  device_setup();
  update->setupflag = 0;
}

void VerletUCGGPU::force_clear() {}

void VerletUCGGPU::run(int n)
{
  int left = n;
  while (left > 0) {
    // up to the next step with thermo / dump / restart output; energy and virial are evaluated on that step only
    const bigint gap = output->next - update->ntimestep;
    const int chunk = (gap > 0 && gap < (bigint) left) ? (int) gap : left;
    const bool out = (update->ntimestep + chunk == output->next);
    if (out) ev_set(update->ntimestep + chunk);    // the step whose energy / virial the output's computes will ask for
    check(ucg_md_run_until(ctx, chunk, out ? 1 : 0));
    update->ntimestep += chunk;
    left -= chunk;
    if (out) {
      download_atoms(true);
      timer->stamp();
      output->write(update->ntimestep);
      timer->stamp(Timer::OUTPUT);
    }
  }
}

void VerletUCGGPU::cleanup()
{
  download_atoms(false);
  modify->post_run();
  domain->box_too_small_check();
  update->update_time();
}

void VerletUCGGPU::reset_dt()
{
  if (ctx) check(ucg_ctx_set_units(ctx, force->boltz, force->ftm2v, force->mvv2e, update->dt, force->special_lj));
}
