// ucg_cluster.hip -- fix cluster_switch (UCG/fix_cluster_switch.cpp) on resident beads (gfx950).
//
// Every switchFreq steps the reference (pre_exchange :452-469) re-neighbours, finds the cluster of
// molecules in contact with molecule mol_seed (check_cluster :551-719: minimum-label propagation
// over a FULL neighbour list, "offset partner" molecules tied in) and lets every molecule outside
// that cluster switch its atoms between the ON and OFF atom types with probability probON /
// probOFF (attempt_switch :721-802, one RanPark draw per deciding molecule in ascending id).
//
//   device : k_cs_sweep     one lane per owned bead walks its row of the full list; for a contact
//                           (allowed type pair, rsq < cutoff^2) between molecules with different
//                           labels it lowers the labels of both molecules and of their partners to
//                           the minimum of the four (atomicMin) -- the reference's update rule
//                           (:619-646), applied until a whole sweep changes nothing.  Labels only
//                           decrease and the stopping condition is the reference's, so the fixed
//                           point is the same as its sequential sweeps' (tested against the oracle).
//            k_cs_molsum    confirm_molecule's sumState (:804-857): +1 per ON-type atom, -1 per OFF
//            k_cs_apply     the type flips of accepted molecules (:762-783)
//   host   : the per-molecule arrays (maxmol+1 ints), the files, RanPark and the decisions --
//            O(molecules) work between two kernels, every switchFreq steps.
//
// Differences from the reference, all in corners it leaves to the local index order of atoms:
//  * a molecule's initial state is taken from its switchable atom with the smallest tag (:140-156 take
//    the first in local order);
//  * confirm_molecule fills at most nSwitchPerMol slots per molecule in local order (:824-849); here every
//    switchable atom of the molecule counts and is switched (identical unless a molecule has more
//    switchable atoms than molecule mol_seed);
//  * partner indices outside 0..maxmol are ignored (the reference reads out of bounds, :629-646);
//  * decomposed runs: the caller performs the reference's MPI_Allreduce steps between the phases exported
//    below (survey scalars and arrays after _create, labels between sweeps, accept flags); as in the
//    reference, the rank holding the majority of a molecule's switchable atoms draws for it from its own
//    RanPark stream, so WHICH molecules switch depends on the decomposition; the cluster labels do not.
//  * the debug log files are not written.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ucg_hip.h"
#include "ucg_ctx.h"

namespace ucg {

struct ClusterSwitch {
  int mol_seed = 0, mol_offset = 0, switchFreq = 0, groupbit = 1;
  double cutsq = 0;
  int ranpark_equal = 0, ranpark_unequal = 0;  // RanPark states (upstream random_park.cpp)
  double probON = 0, probOFF = 0;
  std::vector<int> typesON, typesOFF, contact;  // contact: pairs (itype, jtype)
  int maxmol = -1, nmol = 0, nSwitchPerMol = 0;
  std::vector<int> mol_restrict, mol_state, mol_accept, mol_cluster, present;
  long long next_reneighbor = 0;
  double stats[6] = {0, 0, 0, 0, 0, 0}, nCluster = 0;
  int sweeps = 0;
  long long survey[3] = {0, 0, 0};  // this rank's maxmol, switchable atoms of mol_seed, switchable atoms
  bool synced = true;               // decomposed runs: the survey has been reduced over the ranks
  DevBuf<int> d_lab, d_state, d_accept, d_sum, d_flag, d_typeflag, d_contact;
  // the contacts of the current check_cluster (positions, types and the list are fixed while the labels propagate):
  // per owned bead the molecules it touches, found by the first sweep, walked by the following ones
  DevBuf<int> d_ccount, d_clist;
  bool contacts_valid = false, contacts_overflow = false;
};

void cluster_destroy(ucg_ctx *ctx)
{
  delete ctx->cs;
  ctx->cs = nullptr;
}

namespace {

constexpr int CB = 256;
inline int nblk(long long n) { return (int) ((n + CB - 1) / CB); }

// RanPark::uniform() (upstream random_park.cpp: Park-Miller minimal standard generator)
double ranpark_uniform(int &seed)
{
  const int k = seed / 127773;
  seed = 16807 * (seed - k * 127773) - 2836 * k;
  if (seed < 0) seed += 2147483647;
  return (1.0 / 2147483647.0) * seed;
}

std::vector<std::string> words_of(char *line)
{
  if (char *p = std::strchr(line, '#')) *p = '\0';
  std::vector<std::string> w;
  for (char *t = std::strtok(line, " \t\n\r\f"); t; t = std::strtok(nullptr, " \t\n\r\f")) w.emplace_back(t);
  return w;
}

// read_file (:207-277): the line number counts every physical line
void read_rates(ClusterSwitch &C, const char *file, int ntypes)
{
  FILE *fp = std::fopen(file, "r");
  if (!fp) throw InputError{std::string("Cannot open file ") + file};
  char line[1024];
  int lineNum = 0, nst = 0;
  while (std::fgets(line, sizeof line, fp)) {
    lineNum++;
    const auto w = words_of(line);
    if (w.empty()) continue;
    if (lineNum == 1) {
      C.probON = std::atof(w[0].c_str());
      if (C.probON > 1.0) {
        std::fclose(fp);
        throw InputError{"Incorrect probability in rates.txt files (fix cluster_switch)"};
      }
      C.probOFF = 1.0 - C.probON;
    } else if (lineNum == 2) {
      nst = std::atoi(w[0].c_str());
      if (nst > ntypes || nst < 1) {
        std::fclose(fp);
        throw InputError{"Incorrect number of atom switching types (fix cluster_switch)"};
      }
      C.typesON.assign((size_t) nst, 0);
      C.typesOFF.assign((size_t) nst, 0);
    } else if (lineNum == 3) {
      for (int i = 0; i < nst && i < (int) w.size(); i++) C.typesON[(size_t) i] = std::atoi(w[(size_t) i].c_str());
    } else if (lineNum == 4) {
      for (int i = 0; i < nst && i < (int) w.size(); i++) C.typesOFF[(size_t) i] = std::atoi(w[(size_t) i].c_str());
    }
  }
  std::fclose(fp);
  if (nst == 0) throw InputError{"rates file has no switching types (fix cluster_switch)"};
}

// read_contacts (:281-344)
void read_contacts(ClusterSwitch &C, const char *file)
{
  FILE *fp = std::fopen(file, "r");
  if (!fp) throw InputError{std::string("Cannot open file ") + file};
  char line[1024];
  int lineNum = 0, nct = 0, napc = 0;
  while (std::fgets(line, sizeof line, fp)) {
    lineNum++;
    const auto w = words_of(line);
    if (w.empty()) continue;
    if (lineNum == 1) {
      if (w.size() < 2) { std::fclose(fp); throw InputError{"contacts file: line 1 needs a label and the number of contact types"}; }
      nct = std::atoi(w[1].c_str());
    } else if (lineNum == 2) {
      if (w.size() < 2) { std::fclose(fp); throw InputError{"contacts file: line 2 needs a label and the atoms per contact"}; }
      napc = std::atoi(w[1].c_str());
      if (nct < 1 || napc < 1) { std::fclose(fp); throw InputError{"contacts file: empty contact map"}; }
      C.contact.assign((size_t) nct * napc * 2, 0);
    } else if (!C.contact.empty()) {
      const int off = lineNum - 3;
      if (off >= nct * napc || w.size() < 2) { std::fclose(fp); throw InputError{"contacts file: more pairs than declared"}; }
      C.contact[(size_t) off * 2] = std::atoi(w[0].c_str());
      C.contact[(size_t) off * 2 + 1] = std::atoi(w[1].c_str());
    }
  }
  std::fclose(fp);
  if (C.contact.empty()) throw InputError{"contacts file has no contact map"};
}

bool switchable(const ClusterSwitch &C, int m) { return C.mol_state[(size_t) m] == 0 || C.mol_state[(size_t) m] == 1; }

void check_arrays(const ClusterSwitch &C)
{
  for (int i = 0; i <= C.maxmol; i++)
    if (C.mol_restrict[(size_t) i] == 1 && !switchable(C, i))
      throw InputError{"Communication of mol_state inconsistent: fix cluster_switch"};
}

// ---------------------------------------------------------------------------------- kernels

struct SweepArgs {
  int nlocal, pitch, groupbit, maxmol, mol_offset, ntypes1;
  double cutsq;
};

__device__ __forceinline__ int cs_partner(const int *state, int m, int maxmol, int off)
{
  const int s = state[m];
  const int p = (s == 0 || s == 1) ? m - off : m + off;
  return (p < 0 || p > maxmol) ? -1 : p;
}

// typeflag[itype * ntypes1 + jtype] != 0: (itype, jtype) is in the contact map
__global__ __launch_bounds__(CB) void k_cs_sweep(const SweepArgs S, const double4 *pos4, const int *meta, const int *mask,
                                                const int *mol, const int *numneigh,
                                                const int *neigh, const int *typeflag, const int *state, int *lab,
                                                int *changed)
{
  const int k = blockIdx.x * CB + threadIdx.x;
  if (k >= S.nlocal) return;
  if (!(mask[k] & S.groupbit)) return;
  const int im = mol[k], itype = meta[k] & 0xFFFF;
  const double4 pk = pos4[k];
  const int n = numneigh[k];
  bool any = false;
  for (int e = 0; e < n; e++) {
    const int j = neigh[(size_t) e * S.pitch + k] & 0x1FFFFFFF;
    if (!(mask[j] & S.groupbit)) continue;  // ghosts carry their owner's group bits and molecule id
    const int jm = mol[j];
    const int li = lab[im], lj = lab[jm];
    if (li == lj) continue;
    const int jtype = meta[j] & 0xFFFF;
    if (!typeflag[itype * S.ntypes1 + jtype]) continue;
    const double4 pj = pos4[j];
    const double dx = pk.x - pj.x, dy = pk.y - pj.y, dz = pk.z - pj.z;
    const double rsq = dx * dx + dy * dy + dz * dz;
    if (rsq < S.cutsq) {
      const int pi = cs_partner(state, im, S.maxmol, S.mol_offset), pjm = cs_partner(state, jm, S.maxmol, S.mol_offset);
      int id = min(li, lj);
      if (pi >= 0) id = min(lab[pi], id);
      if (pjm >= 0) id = min(lab[pjm], id);
      bool ch = atomicMin(&lab[im], id) > id;
      ch |= atomicMin(&lab[jm], id) > id;
      if (pi >= 0) ch |= atomicMin(&lab[pi], id) > id;
      if (pjm >= 0) ch |= atomicMin(&lab[pjm], id) > id;
      any |= ch;
    }
  }
  if (any) atomicOr(changed, 1);
}

// The contacts do not change while the labels propagate: the first sweep of a check_cluster also writes, per owned
// bead, the molecules it is in contact with (allowed type pair, inside the cutoff, another molecule) -- about 6 of the
// 70 entries of a row at rho* = 0.8 and cutoff 1.2 -- and the following sweeps (typically a dozen) walk those.  Same
// update rule on the same set of contacts; a row with more than CS_CONTACT_CAP of them keeps the full sweeps.
constexpr int CS_CONTACT_CAP = 24;

template <bool RECORD>
__global__ __launch_bounds__(CB) void k_cs_sweep_contacts(const SweepArgs S, const double4 *pos4, const int *meta, const int *mask,
                                                         const int *mol, const int *numneigh, const int *neigh,
                                                         const int *typeflag, const int *state, int *lab, int *changed,
                                                         int *ccount, int *clist, int *overflow)
{
  const int k = blockIdx.x * CB + threadIdx.x;
  if (k >= S.nlocal) return;
  if (!(mask[k] & S.groupbit)) {
    if (RECORD) ccount[k] = 0;
    return;
  }
  const int im = mol[k];
  bool any = false;
  int nc = 0;
  auto update = [&](const int jm) {
    const int li = lab[im], lj = lab[jm];
    if (li == lj) return;
    const int pi = cs_partner(state, im, S.maxmol, S.mol_offset), pjm = cs_partner(state, jm, S.maxmol, S.mol_offset);
    int id = min(li, lj);
    if (pi >= 0) id = min(lab[pi], id);
    if (pjm >= 0) id = min(lab[pjm], id);
    bool ch = atomicMin(&lab[im], id) > id;
    ch |= atomicMin(&lab[jm], id) > id;
    if (pi >= 0) ch |= atomicMin(&lab[pi], id) > id;
    if (pjm >= 0) ch |= atomicMin(&lab[pjm], id) > id;
    any |= ch;
  };
  if (RECORD) {
    const int itype = meta[k] & 0xFFFF;
    const double4 pk = pos4[k];
    const int n = numneigh[k];
    for (int e = 0; e < n; e++) {
      const int j = neigh[(size_t) e * S.pitch + k] & 0x1FFFFFFF;
      if (!(mask[j] & S.groupbit)) continue;
      const int jm = mol[j];
      if (jm == im) continue;  // one molecule, one label
      if (!typeflag[itype * S.ntypes1 + (meta[j] & 0xFFFF)]) continue;
      const double4 pj = pos4[j];
      const double dx = pk.x - pj.x, dy = pk.y - pj.y, dz = pk.z - pj.z;
      if (dx * dx + dy * dy + dz * dz < S.cutsq) {
        if (nc < CS_CONTACT_CAP) clist[(size_t) nc * S.pitch + k] = jm;
        nc++;
        update(jm);
      }
    }
    ccount[k] = nc;
    if (nc > CS_CONTACT_CAP) atomicOr(overflow, 1);
  } else {
    nc = ccount[k];
    for (int c = 0; c < nc; c++) update(clist[(size_t) c * S.pitch + k]);
  }
  if (any) atomicOr(changed, 1);
}

// onflag[type]: +1 per occurrence of the type among the ON types, offflag likewise (an atom type listed
// twice counts twice, as in the reference's loop over k)
__global__ __launch_bounds__(CB) void k_cs_molsum(int nlocal, int groupbit, const int *meta, const int *mask, const int *mol,
                                                 const int *oncount, const int *offcount, int *sum, int *present)
{
  const int i = blockIdx.x * CB + threadIdx.x;
  if (i >= nlocal) return;
  const int t = meta[i] & 0xFFFF;
  const int d = oncount[t] - offcount[t];
  if (d != 0) atomicAdd(&sum[mol[i]], d);      // confirm_molecule looks at every local atom of the molecule
  if (mask[i] & groupbit) present[mol[i]] = 1;  // the molecules this rank iterates over (:731-739)
}

__global__ __launch_bounds__(CB) void k_cs_apply(int nlocal, int *meta, const int *mol, const int *accept,
                                                const int *state_before, int nst, const int *typesON,
                                                const int *typesOFF)
{
  const int i = blockIdx.x * CB + threadIdx.x;
  if (i >= nlocal) return;
  const int m = mol[i];
  if (accept[m] != 1) return;
  const int mt = meta[i];
  int t = mt & 0xFFFF;
  bool sw = false;
  for (int k = 0; k < nst; k++) sw |= (t == typesON[k] || t == typesOFF[k]);
  if (!sw) return;
  if (state_before[m] == 0) {
    for (int k = 0; k < nst; k++)
      if (t == typesOFF[k]) t = typesON[k];
  } else if (state_before[m] == 1) {
    for (int k = 0; k < nst; k++)
      if (t == typesON[k]) t = typesOFF[k];
  }
  meta[i] = (mt & ~0xFFFF) | t;
}

template <typename T>
void upload(ucg_ctx *ctx, DevBuf<T> &d, const std::vector<T> &h)
{
  d.reserve(h.size() + 1);
  if (!h.empty()) UCG_HIP(hipMemcpyAsync(d.get(), h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
}

template <typename T>
void download(ucg_ctx *ctx, std::vector<T> &h, const T *d, size_t n)
{
  h.resize(n);
  if (n) UCG_HIP(hipMemcpyAsync(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
}

template <typename F>
int guarded(ucg_ctx *ctx, F &&fn)
{
  try {
    (void) hipSetDevice(ctx->device);
    return fn();
  } catch (const InputError &e) {
    ctx->err = e.msg;
    return UCG_ERR_INPUT;
  } catch (const HipFailure &e) {
    ctx->err = std::string("HIP error: ") + hipGetErrorString(e.code) + " in " + e.what;
    return UCG_ERR_HIP;
  } catch (const std::exception &e) {
    ctx->err = e.what();
    return UCG_ERR_INVALID;
  }
}

void need(ucg_ctx *ctx)
{
  if (!ctx->cs) throw InputError{"fix cluster_switch not created"};
  if (!ctx->cs->synced) throw InputError{"fix cluster_switch: the survey of the ranks has not been reduced (ucg_fix_cluster_switch_set_scalars)"};
  if (!ctx->has_mol) throw InputError{"fix cluster_switch requires that atoms have molecule attributes"};
}

}  // namespace

// check_cluster (:551-719) in three phases, so that a decomposed run can reduce the labels over the
// ranks between sweeps (the reference's MPI_Allreduce MIN of mol_cluster, :664):
//   labels_init   the starting labels (:573-599) from the molecules present in the group (anywhere)
//   sweep_local   kernel sweeps over this rank's rows until one changes nothing; returns "changed at all"
//   finalize      restrict / state flags of the seed's cluster (:675-690)
void cluster_labels_init(ucg_ctx *ctx)
{
  need(ctx);
  ClusterSwitch &C = *ctx->cs;
  const int maxmol = C.maxmol;
  std::vector<int> lab((size_t) maxmol + 1, -1);
  lab[(size_t) C.mol_seed] = C.mol_seed;
  lab[(size_t) (C.mol_seed - C.mol_offset)] = C.mol_seed;
  for (int m = 0; m <= maxmol; m++)
    if (C.present[(size_t) m]) lab[(size_t) m] = m;
  for (int m = 0; m <= maxmol; m++)
    if (C.present[(size_t) m] && switchable(C, m)) {
      const int p = m - C.mol_offset;
      if (p >= 0 && p <= maxmol) lab[(size_t) p] = m;
    }
  upload(ctx, C.d_lab, lab);
  upload(ctx, C.d_state, C.mol_state);
  C.sweeps = 0;
  C.contacts_valid = false;  // a new check_cluster: other positions, types, rows
}

bool cluster_sweep_local(ucg_ctx *ctx)
{
  need(ctx);
  ClusterSwitch &C = *ctx->cs;
  if (ctx->list_inum != ctx->nlocal)
    throw InputError{"fix cluster_switch needs the device-built full list (ucg_neigh_rebuild)"};
  C.d_flag.reserve(4);
  SweepArgs S;
  S.nlocal = ctx->nlocal;
  S.pitch = ctx->list_pitch;
  S.groupbit = C.groupbit;
  S.maxmol = C.maxmol;
  S.mol_offset = C.mol_offset;
  S.ntypes1 = ctx->ntypes + 1;
  S.cutsq = C.cutsq;
  bool any = false;
  if (!C.contacts_valid) {
    C.d_ccount.reserve((size_t) S.pitch + 1);
    C.d_clist.reserve((size_t) S.pitch * CS_CONTACT_CAP + 1);
  }
  for (;;) {
    UCG_HIP(hipMemsetAsync(C.d_flag.get(), 0, 2 * sizeof(int), ctx->stream));
    if (ctx->nlocal > 0) {
      if (!C.contacts_valid)  // first sweep of this check_cluster: full rows, contacts recorded
        hipLaunchKernelGGL(k_cs_sweep_contacts<true>, dim3(nblk(ctx->nlocal)), dim3(CB), 0, ctx->stream, S, ctx->pos4.get(),
                           ctx->meta.get(), ctx->mask.get(), ctx->mol.get(), ctx->numneigh.get(), ctx->neigh.get(),
                           C.d_typeflag.get(), C.d_state.get(), C.d_lab.get(), C.d_flag.get(), C.d_ccount.get(),
                           C.d_clist.get(), C.d_flag.get() + 1);
      else if (!C.contacts_overflow)
        hipLaunchKernelGGL(k_cs_sweep_contacts<false>, dim3(nblk(ctx->nlocal)), dim3(CB), 0, ctx->stream, S, ctx->pos4.get(),
                           ctx->meta.get(), ctx->mask.get(), ctx->mol.get(), ctx->numneigh.get(), ctx->neigh.get(),
                           C.d_typeflag.get(), C.d_state.get(), C.d_lab.get(), C.d_flag.get(), C.d_ccount.get(),
                           C.d_clist.get(), C.d_flag.get() + 1);
      else
        hipLaunchKernelGGL(k_cs_sweep, dim3(nblk(ctx->nlocal)), dim3(CB), 0, ctx->stream, S, ctx->pos4.get(), ctx->meta.get(),
                           ctx->mask.get(), ctx->mol.get(), ctx->numneigh.get(), ctx->neigh.get(), C.d_typeflag.get(),
                           C.d_state.get(), C.d_lab.get(), C.d_flag.get());
    }
    UCG_HIP(hipGetLastError());
    int flags[2] = {0, 0};
    UCG_HIP(hipMemcpyAsync(flags, C.d_flag.get(), sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
    UCG_HIP(hipStreamSynchronize(ctx->stream));
    const int changed = flags[0];
    if (!C.contacts_valid) {
      C.contacts_valid = true;
      C.contacts_overflow = flags[1] != 0;
    }
    C.sweeps++;
    if (!changed) break;
    any = true;
    if (C.sweeps > 100000) throw InputError{"fix cluster_switch: label propagation does not converge"};
  }
  return any;
}

void cluster_finalize(ucg_ctx *ctx)
{
  need(ctx);
  ClusterSwitch &C = *ctx->cs;
  const int maxmol = C.maxmol;
  download(ctx, C.mol_cluster, C.d_lab.get(), (size_t) maxmol + 1);
  const int clusterID = C.mol_cluster[(size_t) C.mol_seed];
  C.nCluster = 0.0;
  for (int i = 0; i <= maxmol; i++) {
    if (C.mol_cluster[(size_t) i] == -1) continue;
    if (switchable(C, i)) {
      if (C.mol_cluster[(size_t) i] == clusterID) {
        C.mol_restrict[(size_t) i] = -1;
        C.mol_state[(size_t) i] = 1;
      } else
        C.mol_restrict[(size_t) i] = 1;
    }
    if (C.mol_cluster[(size_t) i] == clusterID) C.nCluster += 1.0;
  }
}

void cluster_check(ucg_ctx *ctx)
{
  cluster_labels_init(ctx);
  cluster_sweep_local(ctx);
  cluster_finalize(ctx);
}

// attempt_switch (:721-802) in two phases: attempt_local decides the molecules this rank is the decision
// maker of (confirm_molecule's majority rule over ITS atoms, one RanPark draw each, ascending id) and leaves
// them in mol_accept; a decomposed run takes the maximum over the ranks (:750) before attempt_apply.
void cluster_attempt_local(ucg_ctx *ctx)
{
  need(ctx);
  ClusterSwitch &C = *ctx->cs;
  const int maxmol = C.maxmol, n = ctx->nlocal;
  const size_t nm = (size_t) maxmol + 1;
  std::vector<int> oncount((size_t) ctx->ntypes + 1, 0), offcount((size_t) ctx->ntypes + 1, 0);
  for (size_t k = 0; k < C.typesON.size(); k++) {
    if (C.typesON[k] >= 0 && C.typesON[k] <= ctx->ntypes) oncount[(size_t) C.typesON[k]]++;
    if (C.typesOFF[k] >= 0 && C.typesOFF[k] <= ctx->ntypes) offcount[(size_t) C.typesOFF[k]]++;
  }
  // an atom matching ON[k] is not tested against OFF[k] for the same k (else-if, :817-846)
  for (size_t k = 0; k < C.typesON.size(); k++)
    if (C.typesON[k] == C.typesOFF[k] && C.typesOFF[k] >= 0 && C.typesOFF[k] <= ctx->ntypes) offcount[(size_t) C.typesOFF[k]]--;
  DevBuf<int> d_on, d_off, d_present;
  upload(ctx, d_on, oncount);
  upload(ctx, d_off, offcount);
  C.d_sum.reserve(nm);
  d_present.reserve(nm);
  UCG_HIP(hipMemsetAsync(C.d_sum.get(), 0, nm * sizeof(int), ctx->stream));
  UCG_HIP(hipMemsetAsync(d_present.get(), 0, nm * sizeof(int), ctx->stream));
  if (n > 0)
    hipLaunchKernelGGL(k_cs_molsum, dim3(nblk(n)), dim3(CB), 0, ctx->stream, n, C.groupbit, ctx->meta.get(), ctx->mask.get(),
                       ctx->mol.get(), d_on.get(), d_off.get(), C.d_sum.get(), d_present.get());
  std::vector<int> sum, here;
  download(ctx, sum, C.d_sum.get(), nm);
  download(ctx, here, d_present.get(), nm);  // the molecules with an in-group atom on THIS rank (the std::map of :731-739)

  const double decisionBuffer = (double) C.nSwitchPerMol / 2.0 - 1.0 + 0.01;
  C.mol_accept.assign(nm, -1);
  for (int mID = 0; mID <= maxmol; mID++) {  // std::map iteration: ascending molecule id
    if (!here[(size_t) mID]) continue;
    int confirmflag = 0;
    if (C.mol_restrict[(size_t) mID] == 1) {
      const double sumState = (double) sum[(size_t) mID];
      if (sumState < (decisionBuffer * -1)) confirmflag = -1;
      else if (sumState > decisionBuffer) confirmflag = 1;
    }
    if (confirmflag != 0) {
      const double checkProb = (C.mol_state[(size_t) mID] == 0) ? C.probON : C.probOFF;  // switch_flag :860-885
      const double r = ranpark_uniform(C.ranpark_unequal);
      C.mol_accept[(size_t) mID] = (r < checkProb) ? 1 : 0;
    }
  }
}

void cluster_attempt_apply(ucg_ctx *ctx)
{
  need(ctx);
  ClusterSwitch &C = *ctx->cs;
  const int maxmol = C.maxmol, n = ctx->nlocal;
  // gather_statistics (:899-935), before the states flip
  for (int i = 0; i <= maxmol; i++) {
    if (C.mol_restrict[(size_t) i] != 1) continue;
    C.stats[0] += 1.0;
    if (C.mol_state[(size_t) i] == 0) {
      C.stats[2] += 1.0;
      if (C.mol_accept[(size_t) i] == 1) { C.stats[1] += 1.0; C.stats[4] += 1.0; }
    } else if (C.mol_state[(size_t) i] == 1) {
      C.stats[3] += 1.0;
      if (C.mol_accept[(size_t) i] == 1) { C.stats[1] += 1.0; C.stats[5] += 1.0; }
    }
  }
  check_arrays(C);
  // the type flips on the device, then the bookkeeping
  upload(ctx, C.d_accept, C.mol_accept);
  upload(ctx, C.d_state, C.mol_state);
  DevBuf<int> d_ton, d_toff;
  upload(ctx, d_ton, C.typesON);
  upload(ctx, d_toff, C.typesOFF);
  if (n > 0)
    hipLaunchKernelGGL(k_cs_apply, dim3(nblk(n)), dim3(CB), 0, ctx->stream, n, ctx->meta.get(), ctx->mol.get(),
                       C.d_accept.get(), C.d_state.get(), (int) C.typesON.size(), d_ton.get(), d_toff.get());
  UCG_HIP(hipGetLastError());
  UCG_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i <= maxmol; i++)
    if (C.mol_accept[(size_t) i] == 1) {
      if (C.mol_state[(size_t) i] == 0) C.mol_state[(size_t) i] = 1;
      else if (C.mol_state[(size_t) i] == 1) C.mol_state[(size_t) i] = 0;
    }
}

void cluster_attempt(ucg_ctx *ctx)
{
  cluster_attempt_local(ctx);
  cluster_attempt_apply(ctx);
}

// the step hook of the resident loop: is a rebuild forced at this step (Neighbor::decide), and the
// pre_exchange work once the lists are fresh
bool cluster_forces_rebuild(const ucg_ctx *ctx) { return ctx->cs && ctx->cs->next_reneighbor == ctx->ntimestep; }

void cluster_pre_exchange(ucg_ctx *ctx)
{
  ClusterSwitch &C = *ctx->cs;
  if (C.switchFreq == 0 || C.next_reneighbor != ctx->ntimestep) return;
  cluster_check(ctx);
  cluster_attempt(ctx);
  C.next_reneighbor = ctx->ntimestep + C.switchFreq;
}

}  // namespace ucg

using namespace ucg;

extern "C" {

int ucg_atoms_upload_molecule(ucg_ctx *ctx, const int *molecule)
{
  if (!ctx || !molecule) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    const size_t n = (size_t) ctx->nlocal;
    ctx->mol.reserve(n + 1);
    if (n) UCG_HIP(hipMemcpyAsync(ctx->mol.get(), molecule, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    UCG_HIP(hipStreamSynchronize(ctx->stream));
    ctx->has_mol = true;
    return UCG_OK;
  });
}

int ucg_atoms_download_molecule(ucg_ctx *ctx, int *molecule)
{
  if (!ctx || !molecule) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (!ctx->has_mol) throw InputError{"no molecule ids were uploaded"};
    if (ctx->nlocal)
      UCG_HIP(hipMemcpyAsync(molecule, ctx->mol.get(), (size_t) ctx->nlocal * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    UCG_HIP(hipStreamSynchronize(ctx->stream));
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_create(ucg_ctx *ctx, int groupbit, int mol_seed, int mol_offset, double cutoff, int seed,
                                  int switch_freq, const char *rate_file, const char *contact_file)
{
  if (!ctx || !rate_file || !contact_file) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (!ctx->has_mol) throw InputError{"fix cluster_switch requires that atoms have molecule attributes"};
    const bool multi = ctx->dom_world > 1;  // the caller then reduces the survey over the ranks (set_scalars / set_array)
    cluster_destroy(ctx);
    ctx->cs = new ClusterSwitch();
    ClusterSwitch &C = *ctx->cs;
    C.mol_seed = mol_seed;
    C.mol_offset = mol_offset;
    C.cutsq = cutoff * cutoff;
    C.switchFreq = switch_freq;
    C.groupbit = groupbit;
    C.ranpark_equal = C.ranpark_unequal = seed;
    C.next_reneighbor = ctx->ntimestep + 1;  // :68
    read_rates(C, rate_file, ctx->ntypes);
    read_contacts(C, contact_file);

    // the constructor's survey of the owned atoms (:96-156), in ascending tag order
    const int n = ctx->nlocal;
    std::vector<int> meta, mol, tag, mask;
    download(ctx, meta, ctx->meta.get(), (size_t) n);
    download(ctx, mol, ctx->mol.get(), (size_t) n);
    download(ctx, tag, ctx->tag.get(), (size_t) n);
    download(ctx, mask, ctx->mask.get(), (size_t) n);
    int nmolatoms = 0, maxmol = -1, nspm = 0;
    const int nst = (int) C.typesON.size();
    for (int i = 0; i < n; i++) {
      if (!(mask[(size_t) i] & groupbit)) continue;
      if (mol[(size_t) i] < 0) throw InputError{"fix cluster_switch: negative molecule id"};
      maxmol = std::max(maxmol, mol[(size_t) i]);
      const int t = meta[(size_t) i] & 0xFFFF;
      for (int j = 0; j < nst; j++)
        if (t == C.typesON[(size_t) j] || t == C.typesOFF[(size_t) j]) {
          nmolatoms++;
          if (mol[(size_t) i] == mol_seed) nspm++;
        }
    }
    if (!multi) {
      if (maxmol < 0) throw InputError{"Selected group does not have any mols (fix cluster_switch)"};
      if (nspm < 1) throw InputError{"fix cluster_switch: molecule mol_seed has no switchable atoms (division by zero in the reference)"};
      if (mol_seed < 0 || mol_seed > maxmol || mol_seed - mol_offset < 0 || mol_seed - mol_offset > maxmol)
        throw InputError{"fix cluster_switch: mol_seed / mol_seed - mol_offset outside 0..maxmol (out-of-bounds write in the reference)"};
    }
    for (int i = 0; i < n; i++) maxmol = std::max(maxmol, mol[(size_t) i]);  // atoms outside the group index the arrays too
    if (maxmol < 0) maxmol = 0;
    C.maxmol = maxmol;
    C.nSwitchPerMol = nspm;
    C.nmol = nspm > 0 ? nmolatoms / nspm : 0;
    C.survey[0] = maxmol;
    C.survey[1] = nspm;
    C.survey[2] = nmolatoms;
    C.synced = !multi;
    const size_t nm = (size_t) maxmol + 1;
    C.mol_restrict.assign(nm, -1);
    C.mol_state.assign(nm, -1);
    C.mol_accept.assign(nm, -1);
    C.mol_cluster.assign(nm, -1);
    C.present.assign(nm, 0);
    std::vector<int> ord((size_t) n);
    for (int i = 0; i < n; i++) ord[(size_t) i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return tag[(size_t) a] < tag[(size_t) b]; });
    for (int q = 0; q < n; q++) {
      const int i = ord[(size_t) q];
      if (!(mask[(size_t) i] & groupbit)) continue;
      const int molID = mol[(size_t) i], t = meta[(size_t) i] & 0xFFFF;
      C.present[(size_t) molID] = 1;
      for (int j = 0; j < nst; j++) {
        if (t == C.typesON[(size_t) j] && C.mol_state[(size_t) molID] == -1) {
          C.mol_state[(size_t) molID] = 1;
          if (molID != mol_seed && molID != (mol_seed - mol_offset)) C.mol_restrict[(size_t) molID] = 1;
        } else if (t == C.typesOFF[(size_t) j] && C.mol_state[(size_t) molID] == -1) {
          C.mol_state[(size_t) molID] = 0;
          if (molID != mol_seed && molID != (mol_seed - mol_offset)) C.mol_restrict[(size_t) molID] = 1;
        }
      }
    }
    check_arrays(C);
    // contact map as a type x type flag matrix
    const int nt1 = ctx->ntypes + 1;
    std::vector<int> tf((size_t) nt1 * nt1, 0);
    for (size_t m = 0; m + 1 < C.contact.size(); m += 2) {
      const int a = C.contact[m], b = C.contact[m + 1];
      if (a >= 0 && a < nt1 && b >= 0 && b < nt1) tf[(size_t) a * nt1 + b] = 1;
    }
    upload(ctx, C.d_typeflag, tf);
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_check_cluster(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    cluster_check(ctx);
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_attempt_switch(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    cluster_attempt(ctx);
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_maxmol(const ucg_ctx *ctx) { return (ctx && ctx->cs) ? ctx->cs->maxmol : -1; }

int ucg_fix_cluster_switch_array(ucg_ctx *ctx, int which, int *out)
{
  if (!ctx || !out || !ctx->cs || which < 0 || which > 5) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    ClusterSwitch &C = *ctx->cs;
    if (which == 5) {  // the labels as they stand on the device (between sweeps of a decomposed run)
      std::vector<int> lab;
      download(ctx, lab, C.d_lab.get(), (size_t) C.maxmol + 1);
      std::memcpy(out, lab.data(), lab.size() * sizeof(int));
      return UCG_OK;
    }
    const std::vector<int> &v = which == 0 ? C.mol_cluster : which == 1 ? C.mol_state : which == 2 ? C.mol_restrict
                                : which == 3 ? C.mol_accept : C.present;
    std::memcpy(out, v.data(), v.size() * sizeof(int));
    return UCG_OK;
  });
}

/* decomposed runs: the caller reduces over the ranks what the reference reduces with MPI_Allreduce --
 * scalars (:114-120: maxmol MAX, switchable atoms of mol_seed SUM, switchable atoms SUM), then mol_state /
 * mol_restrict / presence (MAX, :157-158), the labels between sweeps (MIN, :664) and mol_accept (MAX, :750) */
int ucg_fix_cluster_switch_scalars(const ucg_ctx *ctx, long long *out3)
{
  if (!ctx || !ctx->cs || !out3) return UCG_ERR_INVALID;
  for (int k = 0; k < 3; k++) out3[k] = ctx->cs->survey[k];
  return UCG_OK;
}

int ucg_fix_cluster_switch_set_scalars(ucg_ctx *ctx, long long maxmol, long long nspm, long long nmolatoms)
{
  if (!ctx || !ctx->cs) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    ClusterSwitch &C = *ctx->cs;
    if (maxmol < C.maxmol) throw InputError{"fix cluster_switch: reduced maxmol below the local one"};
    if (nspm < 1) throw InputError{"fix cluster_switch: molecule mol_seed has no switchable atoms (division by zero in the reference)"};
    if (C.mol_seed < 0 || C.mol_seed > maxmol || C.mol_seed - C.mol_offset < 0 || C.mol_seed - C.mol_offset > maxmol)
      throw InputError{"fix cluster_switch: mol_seed / mol_seed - mol_offset outside 0..maxmol (out-of-bounds write in the reference)"};
    const size_t nm = (size_t) maxmol + 1;
    C.mol_restrict.resize(nm, -1);
    C.mol_state.resize(nm, -1);
    C.mol_accept.resize(nm, -1);
    C.mol_cluster.resize(nm, -1);
    C.present.resize(nm, 0);
    C.maxmol = (int) maxmol;
    C.nSwitchPerMol = (int) nspm;
    C.nmol = (int) (nmolatoms / nspm);
    C.synced = true;
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_set_array(ucg_ctx *ctx, int which, const int *in)
{
  if (!ctx || !in || !ctx->cs || which < 1 || which > 5) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    ClusterSwitch &C = *ctx->cs;
    const size_t nm = (size_t) C.maxmol + 1;
    if (which == 5) {
      std::vector<int> lab(in, in + nm);
      upload(ctx, C.d_lab, lab);
      return UCG_OK;
    }
    std::vector<int> &v = which == 1 ? C.mol_state : which == 2 ? C.mol_restrict : which == 3 ? C.mol_accept : C.present;
    v.assign(in, in + nm);
    if (which == 1 || which == 2) check_arrays(C);
    return UCG_OK;
  });
}

/* phases of check_cluster / attempt_switch (see above); `begin` != 0 starts from fresh labels */
int ucg_fix_cluster_switch_sweep(ucg_ctx *ctx, int begin, int *changed)
{
  if (!ctx || !changed) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (begin) cluster_labels_init(ctx);
    *changed = cluster_sweep_local(ctx) ? 1 : 0;
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_finalize(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    cluster_finalize(ctx);
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_attempt_local(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    cluster_attempt_local(ctx);
    return UCG_OK;
  });
}

int ucg_fix_cluster_switch_attempt_apply(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    cluster_attempt_apply(ctx);
    return UCG_OK;
  });
}

/* is a re-neighbour forced at step ctx->ntimestep (ucg_md_set_timestep) / move on to the next switching step */
int ucg_fix_cluster_switch_due(const ucg_ctx *ctx, int *forced, int *switching)
{
  if (!ctx || !forced || !switching) return UCG_ERR_INVALID;
  *forced = cluster_forces_rebuild(ctx) ? 1 : 0;
  *switching = (*forced && ctx->cs->switchFreq != 0) ? 1 : 0;
  return UCG_OK;
}

int ucg_fix_cluster_switch_advance(ucg_ctx *ctx)
{
  if (!ctx || !ctx->cs) return UCG_ERR_INVALID;
  ctx->cs->next_reneighbor = ctx->ntimestep + ctx->cs->switchFreq;
  return UCG_OK;
}

int ucg_fix_cluster_switch_vector(const ucg_ctx *ctx, double *out7)
{
  if (!ctx || !out7 || !ctx->cs) return UCG_ERR_INVALID;
  for (int k = 0; k < 6; k++) out7[k] = ctx->cs->stats[k];
  out7[6] = ctx->cs->nCluster;
  return UCG_OK;
}

}  // extern "C"
