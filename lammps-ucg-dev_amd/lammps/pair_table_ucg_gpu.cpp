// USER-UCG/GPU pair styles: LAMMPS-side binding of libucg_hip.so -- see pair_table_ucg_gpu.h.
//
// Data flow per force evaluation (drop-in mode: LAMMPS keeps ownership of the host arrays):
//   re-neighbour step : ucg_atoms_upload(all per-atom fields) + ucg_neigh_upload_full(full list
//                       with the orientation bit tag[i] <= tag[j] set on every entry)
//   every step        : ucg_atoms_upload_comm(x, ucgstate, ucgl, ucgp of owned + ghost atoms)
//                       ucg_pair_compute  ->  ucg_atoms_download(f, ucgforce, scores [, ucgp])
//                       results are ADDED to LAMMPS' arrays (force_clear has zeroed them)
// Resident mode (one rank, every per-step fix of the deck is a USER-UCG/GPU fix): nothing of this moves on an ordinary
// step -- LAMMPS' arrays are bound as host mirrors (ucg_host_bind), the ghosts are refreshed on the device
// (ucg_halo_forward with the image shifts given at the re-neighbour step) and results stay on the device until a fix
// of this package synchronises them (re-neighbour, thermo and dump steps: fix_ucg_gpu.cpp).
// The device kernels gather over a FULL list and leave nothing on ghosts, so newton_pair may
// stay on (the reverse communication then adds zeros) and no fdotr virial is computed.
#include "pair_table_ucg_gpu.h"

#include "atom.h"
#include "atom_vec_ucg_gpu.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "fix.h"
#include "force.h"
#include "memory.h"
#include "modify.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "update.h"
#include "utils.h"

#include "ucg_hip.h"

#include <cstdlib>
#include <cstring>
#include <vector>

using namespace LAMMPS_NS;

// One process per GPU: the rank's device is its node-local rank modulo the visible devices (the launchers' usual variables;
// none set = the current device, i.e. whatever ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES leave as device 0)
static int pick_device()
{
  const int ndev = ucg_device_count();
  if (ndev <= 1) return -1;
  for (const char *var : {"OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID", "LOCAL_RANK"})
    if (const char *v = getenv(var)) return atoi(v) % ndev;
  return -1;
}

PairTableUCGGPU::PairTableUCGGPU(LAMMPS *lmp, int style) : Pair(lmp), ucg_style(style)
{
  no_virial_fdotr_compute = 1;    // the pair virial is returned by the kernel (reference: silently 0)
  if (ucg_ctx_create(pick_device(), &ctx) != UCG_OK)
    error->all(FLERR, "USER-UCG/GPU: no usable HIP device (there is no CPU fallback in this package)");
  check(ucg_pair_create(ctx, style, &gpair), true);
}

PairTableUCGGPU::~PairTableUCGGPU()
{
  if (copymode) return;
  ucg_pair_destroy(gpair);
  ucg_ctx_destroy(ctx);
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
  }
}

void PairTableUCGGPU::check(int rc, bool all)
{
  if (rc == UCG_OK) return;
  if (all) error->all(FLERR, ucg_last_error(ctx));
  else error->one(FLERR, ucg_last_error(ctx));
}

void PairTableUCGGPU::settings(int narg, char **arg)
{
  AtomVecUCG::get(lmp);    // "This pair style requires atom style ucg."
  check(ucg_ctx_set_units(ctx, force->boltz, force->ftm2v, force->mvv2e, update->dt, force->special_lj), true);
  check(ucg_pair_settings(gpair, narg, arg), true);
  // accepted by the library, so arg[0] is one of the four table styles and arg[1] an integer (ucgld.cpp:660-675)
  tabstyle = !strcmp(arg[0], "lookup") ? UCG_LOOKUP : !strcmp(arg[0], "linear") ? UCG_LINEAR :
             !strcmp(arg[0], "spline") ? UCG_SPLINE : UCG_BITMAP;
  tablength = utils::inumeric(FLERR, arg[1], false, lmp);
}

void PairTableUCGGPU::coeff(int narg, char **arg)
{
  if (!allocated) {
    allocated = 1;
    const int n = atom->ntypes + 1;
    memory->create(setflag, n, n, "pair:setflag");
    memory->create(cutsq, n, n, "pair:cutsq");
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) setflag[i][j] = 0;
  }
  check(ucg_pair_coeff(gpair, atom->ntypes, narg, arg), true);
  // the reference sets setflag for the FORMAL types it touched (UCG/pair_table_ucgld.cpp:844-845);
  // formal types span 1..atom->ntypes, and init_one() below reports any pair still missing
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = i; j <= atom->ntypes; j++) setflag[i][j] = 1;
}

void PairTableUCGGPU::init_style()
{
  // full list, ghosts included as neighbours (they are never row owners)
  neighbor->add_request(this, NeighConst::REQ_FULL);
  if (ucg_style == 2) {
    comm_forward = 2;
    if (comm->nprocs == 1 && atom->map_style == Atom::MAP_NONE)
      error->all(FLERR, "USER-UCG/GPU: pair_style table_ucg_bethe_density needs an atom map (atom_modify map yes)");
  }
  // thermostat temperature, found like the reference does (UCG/pair_table_ucgld.cpp:873-881)
  double *pT = nullptr;
  int pdim;
  for (int ifix = 0; ifix < modify->nfix; ifix++) {
    pT = (double *) modify->fix[ifix]->extract("t_target", pdim);
    if (pT) { T = *pT; break; }
  }
  if (!pT) error->all(FLERR, "USER-UCG/GPU pair styles need a fix that exports t_target (e.g. fix ucgld/langevin)");
  check(ucg_ctx_set_units(ctx, force->boltz, force->ftm2v, force->mvv2e, update->dt, force->special_lj), true);
  check(ucg_pair_init(gpair, atom->ntypes, T), true);
  // resident mode: one rank, orthogonal box, and no fix that is not of this package (any other fix may read or write
  // LAMMPS' arrays at a hook this package does not see)
  // neigh_modify as the input deck gave it.  The integrator of this package replaces the delay for the duration of a
  // resident run and restores it in post_run(); should a run have ended without post_run(), the sentinel is not a setting
  if (neighbor->delay != UCG_GPU_DELAY_SENTINEL) {
    user_every = neighbor->every;
    user_delay = neighbor->delay;
    user_check = neighbor->dist_check;
  }
  driver = utils::strmatch(update->integrate_style, "^verlet/ucg/gpu") ? 1 : 0;
  resident = (!driver && comm->nprocs == 1 && !domain->triclinic) ? 1 : 0;
  for (int ifix = 0; ifix < modify->nfix && resident; ifix++) {
    const char *st = modify->fix[ifix]->style;
    if (strcmp(st, "nve/ucgld") && strcmp(st, "nve/ucgld/wall/hard") && strcmp(st, "ucgld/langevin") && strcmp(st, "ucgstate")) resident = 0;
  }
  last_ncalls = last_lastcall = -1;    // whatever `run` this is, the first compute() uploads
}

void PairTableUCGGPU::bind_mirrors()
{
  auto avec = AtomVecUCG::get(lmp);
  check(ucg_host_bind(ctx, &atom->x[0][0], &atom->v[0][0], &atom->f[0][0], avec->ucgstate, avec->num_ucgstates, avec->ucgl,
                      avec->ucgvl, avec->ucgp, avec->ucgforce, &avec->ucgsoftmaxscores[0][0]), false);
}

double PairTableUCGGPU::init_one(int i, int j)
{
  const double cut = ucg_pair_cut(gpair, i, j);
  if (cut < 0.0) error->all(FLERR, "All pair coeffs are not set");
  return cut;
}

void PairTableUCGGPU::upload_list()
{
  auto avec = AtomVecUCG::get(lmp);
  const int nlocal = atom->nlocal, nghost = atom->nghost;
  check(ucg_atoms_upload(ctx, nlocal, nghost, atom->ntypes, &atom->x[0][0], atom->v ? &atom->v[0][0] : nullptr, atom->type,
                         atom->tag, atom->mask, avec->ucgstate, avec->ucgl, avec->ucgvl, avec->ucgml, avec->ucgp,
                         atom->mass), false);
  const int inum = list->inum;
  if (inum != nlocal) error->one(FLERR, "USER-UCG/GPU: full neighbour list must have one row per owned atom");
  std::vector<int> numneigh((size_t) inum);
  std::vector<long long> first((size_t) inum);
  long long total = 0;
  for (int ii = 0; ii < inum; ii++) {
    const int i = list->ilist[ii];
    first[(size_t) i] = total;
    numneigh[(size_t) i] = list->numneigh[i];
    total += list->numneigh[i];
  }
  std::vector<int> flat((size_t) total + 1);
  const tagint *tag = atom->tag;
  for (int ii = 0; ii < inum; ii++) {
    const int i = list->ilist[ii];
    const int *jlist = list->firstneigh[i];
    int *dst = &flat[(size_t) first[(size_t) i]];
    for (int jj = 0; jj < list->numneigh[i]; jj++) {
      const int j = jlist[jj] & NEIGHMASK;
      const int sb = (jlist[jj] >> SBBITS) & 3;
      const int orient = (tag[i] <= tag[j]) ? 1 : 0;    // the row owner plays the reference's "i"
      dst[jj] = j | (orient << UCG_ORIENT_BIT) | (sb << UCG_SBBITS);
    }
  }
  check(ucg_neigh_upload_full(ctx, inum, numneigh.data(), first.data(), flat.data()), false);
  if (ucg_style == 2 && comm->nprocs == 1 && nghost > 0) {
    // one rank: every ghost is a periodic image of an owned atom; the density style gives it its owner's prior and CV
    // force on the device (the forward_comm the reference declares, UCG/pair_table_ucg_bethe_density.cpp:280)
    std::vector<int> src((size_t) nghost);
    for (int g = 0; g < nghost; g++) {
      const int o = atom->map(tag[nlocal + g]);
      if (o < 0 || o >= nlocal) error->one(FLERR, "USER-UCG/GPU: ghost atom without an owned image (atom_modify map needed)");
      src[(size_t) g] = o;
    }
    check(ucg_ghosts_upload(ctx, src.data(), nghost), false);
  }
  if (resident) {
    // every ghost of a single rank is a periodic image: its owner and the box shifts CommBrick applied
    std::vector<int> src((size_t) nghost), sh((size_t) nghost * 3);
    if (nghost > 0 && atom->map_style == Atom::MAP_NONE) error->all(FLERR, "USER-UCG/GPU resident mode needs an atom map (atom_modify map yes)");
    for (int g = 0; g < nghost; g++) {
      const int o = atom->map(tag[nlocal + g]);
      if (o < 0 || o >= nlocal) error->one(FLERR, "USER-UCG/GPU: ghost atom without an owned image");
      src[(size_t) g] = o;
      for (int d = 0; d < 3; d++) {
        const double k = (atom->x[nlocal + g][d] - atom->x[o][d]) / domain->prd[d];
        sh[3 * (size_t) g + d] = k > 0.5 ? 1 : (k < -0.5 ? -1 : 0);
      }
    }
    check(ucg_domain_set(ctx, domain->boxlo, domain->boxhi, ucg_pair_cutforce(gpair), neighbor->skin, user_every, user_delay,
                         user_check), false);
    check(ucg_ghosts_upload_images(ctx, src.data(), sh.data(), nghost), false);
    bind_mirrors();    // (again after every re-neighbouring: grow_pointers may have moved LAMMPS' arrays)
  }
  // a list is stale when Neighbor has built since -- ncalls counts the builds of this run (it restarts at every `run`),
  // lastcall is the timestep of the last one: both are kept, so a second `run` whose setup() build leaves ncalls where
  // the previous run ended is still seen -- or when the atom counts changed
  last_ncalls = neighbor->ncalls;
  last_lastcall = neighbor->lastcall;
  last_nlocal = nlocal;
  last_nghost = nghost;
}

int PairTableUCGGPU::pack_forward_comm(int n, int *list, double *buf, int /*pbc_flag*/, int * /*pbc*/)
{
  int m = 0;
  for (int i = 0; i < n; i++) {
    const int j = list[i];
    buf[m++] = aux[2 * (size_t) j];
    buf[m++] = aux[2 * (size_t) j + 1];
  }
  return m;
}

void PairTableUCGGPU::unpack_forward_comm(int n, int first, double *buf)
{
  int m = 0;
  for (int i = first; i < first + n; i++) {
    aux[2 * (size_t) i] = buf[m++];
    aux[2 * (size_t) i + 1] = buf[m++];
  }
}

void PairTableUCGGPU::compute(int eflag, int vflag)
{
  if (driver) error->all(FLERR, "USER-UCG/GPU: Pair::compute() called under run_style verlet/ucg/gpu, which runs the step loop "
                                "inside the library (a command that needs forces outside a run?)");
  ev_init(eflag, vflag);
  auto avec = AtomVecUCG::get(lmp);
  const int nlocal = atom->nlocal;
  if (neighbor->ncalls != last_ncalls || neighbor->lastcall != last_lastcall || nlocal != last_nlocal || atom->nghost != last_nghost)
    upload_list();
  else if (resident) check(ucg_halo_forward(ctx), false);    // owner -> periodic images on the device; nothing is uploaded
  else check(ucg_atoms_upload_comm(ctx, &atom->x[0][0], avec->ucgstate, avec->ucgl, avec->ucgp), false);

  double eng = 0.0, vir[6] = {0, 0, 0, 0, 0, 0};
  if (ucg_style == 2 && comm->nprocs > 1) {
    // the three passes of compute() one at a time; between them the ghosts' priors (after pass 1) and CV forces
    // (after pass 2) come from their owner ranks through LAMMPS' own forward communication
    const int nall = nlocal + atom->nghost;
    aux.assign(2 * (size_t) nall, 0.0);
    comm_forward = 2;
    for (int phase = 1; phase <= 3; phase++) {
      check(ucg_pair_density_phase(gpair, phase, eflag_global, vflag_global, phase == 3 ? &eng : nullptr,
                                   phase == 3 ? vir : nullptr), false);
      if (phase == 3) break;
      check(ucg_pair_density_aux_download(gpair, phase - 1, aux.data(), 0, nlocal), false);
      comm->forward_comm(this);
      check(ucg_pair_density_aux_upload(gpair, phase - 1, aux.data() + 2 * (size_t) nlocal, nlocal, atom->nghost), false);
    }
  } else {
    check(ucg_pair_compute(gpair, eflag_global, vflag_global, &eng, vir), false);
  }
  check(ucg_pair_check_errors(gpair), false);    // "Pair distance < table inner cutoff" etc.
  if (resident) {
    // f, ucgforce, scores, num_ucgstates (and ucgp) stay on the device: the package's fixes read them there, and the host
    // mirrors catch up at the next ucg_host_sync (LAMMPS' force_clear has zeroed the host arrays: the device values are
    // the whole force in a deck that qualifies for this mode)
    if (eflag_global) eng_vdwl += eng;
    if (vflag_global)
      for (int k = 0; k < 6; k++) virial[k] += vir[k];
    return;
  }

  std::vector<double> f((size_t) nlocal * 3), uf((size_t) nlocal), sc((size_t) nlocal * 2), up;
  std::vector<int> ns((size_t) nlocal);
  if (ucg_style == 2) up.resize((size_t) nlocal);
  check(ucg_atoms_download(ctx, 0, nullptr, nullptr, f.data(), nullptr, nullptr, nullptr, ns.data(), nullptr, nullptr,
                           nullptr, up.empty() ? nullptr : up.data(), uf.data(), sc.data()), false);
  double **af = atom->f;
  for (int i = 0; i < nlocal; i++) {
    af[i][0] += f[3 * (size_t) i];
    af[i][1] += f[3 * (size_t) i + 1];
    af[i][2] += f[3 * (size_t) i + 2];
    avec->num_ucgstates[i] = ns[(size_t) i];
    if (ucg_style == 0) avec->ucgforce[i] += uf[(size_t) i];
    if (ucg_style == 1) {
      // table_ucg_bethe ASSIGNS the chemical-potential term (UCG/pair_table_ucg_bethe.cpp:155-162)
      avec->ucgsoftmaxscores[i][0] = sc[2 * (size_t) i];
      avec->ucgsoftmaxscores[i][1] = sc[2 * (size_t) i + 1];
    } else if (ucg_style == 0) {
      avec->ucgsoftmaxscores[i][0] += sc[2 * (size_t) i];
      avec->ucgsoftmaxscores[i][1] += sc[2 * (size_t) i + 1];
    }
    if (ucg_style == 2) avec->ucgp[i] = up[(size_t) i];
  }
  if (eflag_global) eng_vdwl += eng;
  if (vflag_global)
    for (int k = 0; k < 6; k++) virial[k] += vir[k];
}

double PairTableUCGGPU::single(int, int, int itype, int jtype, double rsq, double, double factor_lj, double &fforce)
{
  double e = 0.0;
  check(ucg_pair_single(gpair, itype, jtype, rsq, factor_lj, &fforce, &e), false);
  return e;
}

// Restart files: exactly what the reference writes (UCG/pair_table_ucgld.cpp:1431-1473) -- the table style, its length
// and the long-range flags; tables, the state-settings file and coefficients are NOT stored: pair_style / pair_coeff
// are given again after read_restart, as with the reference.
void PairTableUCGGPU::write_restart(FILE *fp)
{
  write_restart_settings(fp);
}

void PairTableUCGGPU::read_restart(FILE *fp)
{
  read_restart_settings(fp);
  if (!allocated) {
    allocated = 1;
    const int n = atom->ntypes + 1;
    memory->create(setflag, n, n, "pair:setflag");
    memory->create(cutsq, n, n, "pair:cutsq");
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) setflag[i][j] = 0;
  }
}

void PairTableUCGGPU::write_restart_settings(FILE *fp)
{
  fwrite(&tabstyle, sizeof(int), 1, fp);
  fwrite(&tablength, sizeof(int), 1, fp);
  fwrite(&ewaldflag, sizeof(int), 1, fp);
  fwrite(&pppmflag, sizeof(int), 1, fp);
  fwrite(&msmflag, sizeof(int), 1, fp);
  fwrite(&dispersionflag, sizeof(int), 1, fp);
  fwrite(&tip4pflag, sizeof(int), 1, fp);
}

void PairTableUCGGPU::read_restart_settings(FILE *fp)
{
  int *vals[7] = {&tabstyle, &tablength, &ewaldflag, &pppmflag, &msmflag, &dispersionflag, &tip4pflag};
  if (comm->me == 0)
    for (int *v : vals) utils::sfread(FLERR, v, sizeof(int), 1, fp, nullptr, error);
  for (int *v : vals) MPI_Bcast(v, 1, MPI_INT, 0, world);
}

void *PairTableUCGGPU::extract(const char *str, int &dim)
{
  // the USER-UCG/GPU fixes share this style's device context (fix_ucg_gpu.cpp)
  dim = 0;
  if (strcmp(str, "ucg_ctx") == 0) return (void *) ctx;
  if (strcmp(str, "ucg_resident") == 0) return (void *) &resident;
  if (strcmp(str, "ucg_driver") == 0) return (void *) &driver;
  return nullptr;
}
