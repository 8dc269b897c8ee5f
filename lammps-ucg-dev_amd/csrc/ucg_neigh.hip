// ucg_neigh.hip -- what sits either side of the pair kernel in a resident run:
// periodic wrap, bead sorting, periodic-image ghosts, binning, the full neighbour list,
// the per-step halo refresh, the re-neighbour decision and the Verlet step loop.
//
// None of this is in the reference tree: it is upstream LAMMPS (Domain::pbc, AtomSort,
// CommBrick::borders/forward_comm, NBin/NPair, Neighbor::decide, Verlet::setup/run).
// The step ORDER follows upstream Verlet as summarised in SURVEY.md section 3.1; the data
// management follows the "ucg-rebuild-v1" specification, which the CPU
// oracle implements independently so that whole trajectories can be compared bit for bit:
//   (1) wrap owned beads into the box; (2) bins of ~cutneigh/2 over the box extended by
//   cutneigh; (3) sort owned beads by (Morton code of the bin, tag); (4) ghosts = periodic images
//   that fall in the extended box, sorted by (Morton code of the bin, tag, shift code); (5) full-list rows in stencil order
//   (dz, dy, dx ascending), owned beads of a bin before its ghosts, kept when
//   rsq < cutneigh^2, bit 29 = (tag_row <= tag_neighbour); each row stably partitioned into
//   four build-time distance classes (inside the force cutoff, then thirds of the skin).
// Everything is deterministic: unique sort keys, no order-dependent atomics in outputs.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstring>

#include "../../include/ucg_hip.h"
#include "ucg_ctx.h"

namespace ucg {

struct Domain {
  double boxlo[3], boxhi[3], prd[3];
  // spatial decomposition (1x1x1 = the whole periodic box on one GPU)
  int procgrid[3] = {1, 1, 1}, myloc[3] = {0, 0, 0}, me = 0, world = 1;
  double sublo[3], subhi[3];
  // multi-rank halo: send lists grouped by destination rank, fixed between rebuilds
  DevBuf<int> send_src, send_code, ghost_perm, dest_of, slot_of, offsets, holes, movers;
  DevBuf<char> sendbuf;
  std::vector<long long> send_counts, recv_counts;
  long long nsend = 0;
  double cutforce = 0, skin = 0, cutneigh = 0;
  int every = 1, delay = 0, check = 1;
  int nbin[3] = {1, 1, 1}, sten[3] = {0, 0, 0}, nbins = 1;
  double bboxlo[3], bboxhi[3], binsize[3], bininv[3];
  int ago = 0;
  DevBuf<int> bin_of, ghost_code, counter, rowclass, blockflags, blockoffset;
  DevBuf<int4> cells;     // per bin {owned start, owned end, ghost start, ghost end}
  DevBuf<int4> blockstat; // per brick of the tiled row builder {max row, max skin, entries, -}
  DevBuf<double4> bpos;   // builder records {x, y, z, (double) tag}
  DevBuf<double4> xhold, tmp4;
  DevBuf<unsigned long long> keys_in, keys_out;
  DevBuf<int> vals_in, vals_out, tmpi, cand_src, cand_code;
  DevBuf<double> tmpd;
  DevBuf<char> cub_tmp;
  DevBuf<int> scratch;                 // discovery-order rows before the class partition
  DevBuf<unsigned long long> rowstat;  // [0] max row length, [1] total entries
  int row_capacity = 0;
  // second scratch set of the one-launch permutation of all per-bead arrays (k_permute_all)
  DevBuf<double4> tmp4b;
  DevBuf<int> tmpi2, tmpi3, tmpi4, tmpi5;
  DevBuf<double> tmpd2;
  // pinned staging of the per-destination offsets (two alternating slots: a rebuild uploads them twice, with stream
  // synchronisations of its own in between, so a slot is never rewritten before its copy has run)
  int *h_off[2] = {nullptr, nullptr};
  int h_off_cap = 0, h_off_next = 0;
  ~Domain()
  {
    for (int *p : h_off)
      if (p) (void) hipHostFree(p);
  }
};

struct DomainDev {
  double boxlo[3], boxhi[3], prd[3];
  int procgrid[3], me;
  double bboxlo[3], bboxhi[3], bininv[3];
  int nbin[3], sten[3];
  double cutneighsq, triggersq;
  double cls_sq[3];  // build-time distance classes of a row: force cutoff, then thirds of the skin
};

void domain_destroy(ucg_ctx *ctx)
{
  delete ctx->dom;
  ctx->dom = nullptr;
}

namespace {

constexpr int NB = 256;
inline int nblk(long long n) { return (int) ((n + NB - 1) / NB); }

DomainDev make_dev(const Domain &D)
{
  DomainDev d;
  for (int k = 0; k < 3; k++) {
    d.boxlo[k] = D.boxlo[k];
    d.boxhi[k] = D.boxhi[k];
    d.prd[k] = D.prd[k];
    d.bboxlo[k] = D.bboxlo[k];
    d.bboxhi[k] = D.bboxhi[k];
    d.bininv[k] = D.bininv[k];
    d.nbin[k] = D.nbin[k];
    d.sten[k] = D.sten[k];
    d.procgrid[k] = D.procgrid[k];
  }
  d.me = D.me;
  d.cutneighsq = D.cutneigh * D.cutneigh;
  d.triggersq = 0.25 * D.skin * D.skin;
  for (int c = 0; c < 3; c++) {
    const double rc = D.cutforce + D.skin * c / 3.0;
    d.cls_sq[c] = rc * rc;
  }
  return d;
}

__device__ __forceinline__ int coord2bin(const DomainDev &D, double x, double y, double z)
{
  int bx = (int) ((x - D.bboxlo[0]) * D.bininv[0]);
  int by = (int) ((y - D.bboxlo[1]) * D.bininv[1]);
  int bz = (int) ((z - D.bboxlo[2]) * D.bininv[2]);
  bx = bx < 0 ? 0 : (bx > D.nbin[0] - 1 ? D.nbin[0] - 1 : bx);
  by = by < 0 ? 0 : (by > D.nbin[1] - 1 ? D.nbin[1] - 1 : by);
  bz = bz < 0 ? 0 : (bz > D.nbin[2] - 1 ? D.nbin[2] - 1 : bz);
  return (bz * D.nbin[1] + by) * D.nbin[0] + bx;
}

// Morton (Z-order) code of a bin, 9 bits per dimension: beads sorted along this curve form
// compact blobs, so most neighbours of a workgroup's beads are the workgroup's own beads
__device__ __forceinline__ unsigned long long morton_of_bin(const DomainDev &D, int b)
{
  const unsigned bx = (unsigned) (b % D.nbin[0]), by = (unsigned) ((b / D.nbin[0]) % D.nbin[1]);
  const unsigned bz = (unsigned) (b / (D.nbin[0] * D.nbin[1]));
  unsigned long long m = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    m |= (unsigned long long) ((bx >> i) & 1u) << (3 * i);
    m |= (unsigned long long) ((by >> i) & 1u) << (3 * i + 1);
    m |= (unsigned long long) ((bz >> i) & 1u) << (3 * i + 2);
  }
  return m;
}

__device__ __forceinline__ double wrap1(double x, double lo, double hi, double prd)
{
  // Domain::pbc(), orthogonal periodic
  if (x < lo) x += prd;
  if (x >= hi) {
    x -= prd;
    x = (x > lo) ? x : lo;
  }
  return x;
}

__global__ __launch_bounds__(NB) void k_wrap_and_key(const DomainDev D, int n, double4 *pos4, const int *tag,
                                                    unsigned long long *keys, int *vals)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  double4 p = pos4[i];
  p.x = wrap1(p.x, D.boxlo[0], D.boxhi[0], D.prd[0]);
  p.y = wrap1(p.y, D.boxlo[1], D.boxhi[1], D.prd[1]);
  p.z = wrap1(p.z, D.boxlo[2], D.boxhi[2], D.prd[2]);
  pos4[i] = p;
  const unsigned long long b = morton_of_bin(D, coord2bin(D, p.x, p.y, p.z));
  keys[i] = (b << 37) | ((unsigned long long) (unsigned int) tag[i] << 5);
  vals[i] = i;
}

template <typename T>
__global__ __launch_bounds__(NB) void k_gather(int n, const int *perm, const T *src, T *dst)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}

// every per-bead array of the owned beads through the sort permutation, and the beads' bins, in ONE launch (nine
// k_gather launches and k_bins_from_pos before: at 125 k beads per rank each was 2-3 us of work behind a 4-5 us dispatch)
struct PermuteArgs {
  const double4 *pos_s, *vel_s;
  double4 *pos_d, *vel_d;
  const int *meta_s, *tag_s, *mask_s, *nst_s, *mol_s;
  int *meta_d, *tag_d, *mask_d, *nst_d, *mol_d;
  const double *ml_s, *up_s;
  double *ml_d, *up_d;
};
__global__ __launch_bounds__(NB) void k_permute_all(const DomainDev D, int n, const int *perm, const PermuteArgs a, int *bin_of)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  const int j = perm[i];
  const double4 p = a.pos_s[j];
  a.pos_d[i] = p;
  a.vel_d[i] = a.vel_s[j];
  a.meta_d[i] = a.meta_s[j];
  a.tag_d[i] = a.tag_s[j];
  a.mask_d[i] = a.mask_s[j];
  a.nst_d[i] = a.nst_s[j];
  a.ml_d[i] = a.ml_s[j];
  a.up_d[i] = a.up_s[j];
  if (a.mol_s) a.mol_d[i] = a.mol_s[j];
  bin_of[i] = coord2bin(D, p.x, p.y, p.z);
}

__global__ __launch_bounds__(NB) void k_bins_from_pos(const DomainDev D, int n, int offset, const double4 *pos4, int *bin_of)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  const double4 p = pos4[offset + i];
  bin_of[offset + i] = coord2bin(D, p.x, p.y, p.z);
}

// periodic images of owned beads that fall inside the extended box.  No atomics: FILL=false writes the number of
// images per workgroup, the host scans them, FILL=true writes the images at blockoffset + (wavefront, shift, lane)
// rank -- the same ballots in both passes.  (Thousands of same-address atomics took 50 us per pass.)
template <bool FILL>
__global__ __launch_bounds__(NB) void k_ghost_candidates(const DomainDev D, int n, const double4 *pos4, const int *tag,
                                                        int *blockcount, const int *blockoffset,
                                                        unsigned long long *keys, int *vals, int *cand_src,
                                                        int *cand_code)
{
  __shared__ int s_wave[NB / 64];
  const int i = blockIdx.x * NB + threadIdx.x;
  const bool live = i < n;
  const double4 p = live ? pos4[i] : make_double4(0, 0, 0, 0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // most wavefronts hold interior beads only (Morton order): none of their beads has an image in the shell
  const bool shell = live && (p.x + D.prd[0] < D.bboxhi[0] || p.x - D.prd[0] >= D.bboxlo[0] || p.y + D.prd[1] < D.bboxhi[1] ||
                              p.y - D.prd[1] >= D.bboxlo[1] || p.z + D.prd[2] < D.bboxhi[2] || p.z - D.prd[2] >= D.bboxlo[2]);
  const bool any_shell = __any(shell);
  // pass over the 26 shifts: this wavefront's number of images
  int mine = 0;
  if (any_shell) {
    for (int code = 0; code < 27; code++) {
      if (code == 13) continue;
      const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
      const double xs = p.x + sx * D.prd[0], ys = p.y + sy * D.prd[1], zs = p.z + sz * D.prd[2];
      const bool hit = live && !(zs < D.bboxlo[2] || zs >= D.bboxhi[2]) && !(ys < D.bboxlo[1] || ys >= D.bboxhi[1]) &&
                       !(xs < D.bboxlo[0] || xs >= D.bboxhi[0]);
      mine += __popcll(__ballot(hit));
    }
  }
  if (lane == 0) s_wave[wave] = mine;
  __syncthreads();
  if (!FILL) {
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < NB / 64; w++) tot += s_wave[w];
      blockcount[blockIdx.x] = tot;
    }
    return;
  }
  if (!any_shell) return;
  int base = blockoffset[blockIdx.x];
  for (int w = 0; w < wave; w++) base += s_wave[w];
  for (int code = 0; code < 27; code++) {
    if (code == 13) continue;
    const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
    const double xs = p.x + sx * D.prd[0], ys = p.y + sy * D.prd[1], zs = p.z + sz * D.prd[2];
    const bool hit = live && !(zs < D.bboxlo[2] || zs >= D.bboxhi[2]) && !(ys < D.bboxlo[1] || ys >= D.bboxhi[1]) &&
                     !(xs < D.bboxlo[0] || xs >= D.bboxhi[0]);
    const unsigned long long mask = __ballot(hit);
    if (hit) {
      const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
      const unsigned long long b = morton_of_bin(D, coord2bin(D, xs, ys, zs));
      keys[slot] = (b << 37) | ((unsigned long long) (unsigned int) tag[i] << 5) | (unsigned long long) code;
      vals[slot] = slot;
      cand_src[slot] = i;
      cand_code[slot] = code;
    }
    base += __popcll(mask);
  }
}

__global__ __launch_bounds__(NB) void k_ghost_finalize(int ng, int nlocal, const int *order, const unsigned long long *keys,
                                                      const int *cand_src, const int *cand_code, int *ghost_src,
                                                      int *ghost_code, int *bin_of, int *tag, int *meta)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g >= ng) return;
  const int c = order[g];
  const int src = cand_src[c];
  ghost_src[g] = src;
  ghost_code[g] = cand_code[c];
  tag[nlocal + g] = tag[src];
  meta[nlocal + g] = meta[src];
}

// forward communication owner -> periodic image: fields_comm of UCG/atom_vec_ucg.cpp:71
// (x + shift, ucgstate, ucgl, ucgp)
__global__ __launch_bounds__(NB) void k_halo_forward(const DomainDev D, int ng, int nlocal, const int *ghost_src,
                                                    const int *ghost_code, double4 *pos4, int *meta, double *ucgp)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g >= ng) return;
  const int src = ghost_src[g], code = ghost_code[g];
  const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
  double4 p = pos4[src];
  p.x = p.x + sx * D.prd[0];
  p.y = p.y + sy * D.prd[1];
  p.z = p.z + sz * D.prd[2];
  pos4[nlocal + g] = p;
  meta[nlocal + g] = meta[src];
  ucgp[nlocal + g] = ucgp[src];
}

__global__ __launch_bounds__(NB) void k_ghost_copy_int2(int ng, int nlocal, const int *ghost_src, int *a, int *b)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g >= ng) return;
  const int src = ghost_src[g];
  a[nlocal + g] = a[src];
  b[nlocal + g] = b[src];
}

// per bin {owned start, owned end, ghost start, ghost end}; cls selects the pair to fill
__global__ __launch_bounds__(NB) void k_cell_ranges(int n, int offset, const int *bin_of, int4 *cells, int cls)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  const int b = bin_of[offset + i];
  int *c = reinterpret_cast<int *>(&cells[b]) + 2 * cls;
  if (i == 0 || bin_of[offset + i - 1] != b) c[0] = offset + i;
  if (i == n - 1 || bin_of[offset + i + 1] != b) c[1] = offset + i + 1;
}

// builder records: position + tag in one 32-byte gather
__global__ __launch_bounds__(NB) void k_builder_records(int nall, const double4 *pos4, const int *tag, double4 *bpos)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= nall) return;
  double4 p = pos4[i];
  p.w = (double) tag[i];
  bpos[i] = p;
}

// full-list rows, pass 1 ("discover"): one lane per owned bead walks the stencil bins in
// (dz, dy, dx) ascending order -- bins farther than cutneigh from the bead itself are skipped --
// and appends every neighbour within cutneigh to a scratch row in discovery order, tagged with
// its build-time distance class in bits 30-31 (class 0: inside the force cutoff; 1..3: thirds of
// the skin shell).  Rows longer than `cap` are counted but not stored (the host then grows the
// buffers and repeats).  Per-class counts, row length, and block-reduced max / total are kept.
__global__ __launch_bounds__(NB) void k_rows_discover(const DomainDev D, int nlocal, const double4 *pos4,
                                                     const int *bin_of, const int4 *cells, int *rowcount, int *rowclass,
                                                     int *scratch, int pitch, int cap, const double3 binsize,
                                                     int *maxrow, unsigned long long *total)
{
  __shared__ int s_max[NB / 64];
  __shared__ unsigned long long s_tot[NB / 64];
  const int k = blockIdx.x * NB + threadIdx.x;
  int cnt = 0;
  if (k < nlocal) {
    const double4 pk = pos4[k];  // builder record: w = tag
    const double tk = pk.w;
    const int b = bin_of[k];
    const int bx = b % D.nbin[0], by = (b / D.nbin[0]) % D.nbin[1], bz = b / (D.nbin[0] * D.nbin[1]);
    const double prune = D.cutneighsq * (1.0 + 1.0e-9) + 1.0e-12;
    int c0 = 0, c1 = 0, c2 = 0;
    for (int dz = -D.sten[2]; dz <= D.sten[2]; dz++) {
      const int cz = bz + dz;
      if (cz < 0 || cz >= D.nbin[2]) continue;
      const double zlo = D.bboxlo[2] + cz * binsize.z;
      const double ez = pk.z < zlo ? zlo - pk.z : (pk.z > zlo + binsize.z ? pk.z - (zlo + binsize.z) : 0.0);
      for (int dy = -D.sten[1]; dy <= D.sten[1]; dy++) {
        const int cy = by + dy;
        if (cy < 0 || cy >= D.nbin[1]) continue;
        const double ylo = D.bboxlo[1] + cy * binsize.y;
        const double ey = pk.y < ylo ? ylo - pk.y : (pk.y > ylo + binsize.y ? pk.y - (ylo + binsize.y) : 0.0);
        if (ez * ez + ey * ey > prune) continue;
        for (int dx = -D.sten[0]; dx <= D.sten[0]; dx++) {
          const int cx = bx + dx;
          if (cx < 0 || cx >= D.nbin[0]) continue;
          const double xlo = D.bboxlo[0] + cx * binsize.x;
          const double ex = pk.x < xlo ? xlo - pk.x : (pk.x > xlo + binsize.x ? pk.x - (xlo + binsize.x) : 0.0);
          if (ez * ez + ey * ey + ex * ex > prune) continue;
          const int c = (cz * D.nbin[1] + cy) * D.nbin[0] + cx;
          const int4 cell = cells[c];
          for (int cls = 0; cls < 2; cls++) {
            const int m0 = cls ? cell.z : cell.x;
            const int m1 = cls ? cell.w : cell.y;
            for (int m = m0; m < m1; m++) {
              if (m == k) continue;
              const double4 pm = pos4[m];
              const double delx = pk.x - pm.x;
              const double dely = pk.y - pm.y;
              const double delz = pk.z - pm.z;
              const double rsq = delx * delx + dely * dely + delz * delz;
              if (rsq < D.cutneighsq) {
                int rc;
                if (rsq < D.cls_sq[0]) { rc = 0; c0++; }
                else if (rsq < D.cls_sq[1]) { rc = 1; c1++; }
                else if (rsq < D.cls_sq[2]) { rc = 2; c2++; }
                else rc = 3;
                if (cnt < cap) {
                  const int orient = (tk <= pm.w) ? 1 : 0;
                  scratch[(size_t) cnt * pitch + k] = m | (orient << UCG_ORIENT_BIT) | (rc << 30);
                }
                cnt++;
              }
            }
          }
        }
      }
    }
    rowcount[k] = cnt;
    rowclass[k] = c0;
    rowclass[pitch + k] = c1;
    rowclass[2 * pitch + k] = c2;
  }
  // block max / total -> one atomic each per block (integer: order-free)
  int mx = cnt;
  unsigned long long tot = (unsigned long long) cnt;
  for (int off = 32; off > 0; off >>= 1) {
    mx = max(mx, __shfl_down(mx, off, 64));
    tot += __shfl_down(tot, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_max[threadIdx.x >> 6] = mx;
    s_tot[threadIdx.x >> 6] = tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NB / 64; w++) {
      mx = max(mx, s_max[w]);
      tot += s_tot[w];
    }
    atomicMax(maxrow, mx);
    atomicAdd(total, tot);
  }
}

// pass 2 ("partition"): stable partition of each scratch row into its four classes
__global__ __launch_bounds__(NB) void k_rows_partition(int nlocal, const int *rowcount, const int *rowclass,
                                                      const int *scratch, int *neigh, int pitch)
{
  const int k = blockIdx.x * NB + threadIdx.x;
  if (k >= nlocal) return;
  const int n = rowcount[k];
  int p0 = 0;
  int p1 = rowclass[k];
  int p2 = p1 + rowclass[pitch + k];
  int p3 = p2 + rowclass[2 * pitch + k];
  for (int e = 0; e < n; e++) {
    const int ent = scratch[(size_t) e * pitch + k];
    const int rc = (ent >> 30) & 3;
    int slot;
    if (rc == 0) slot = p0++;
    else if (rc == 1) slot = p1++;
    else if (rc == 2) slot = p2++;
    else slot = p3++;
    neigh[(size_t) slot * pitch + k] = ent & 0x3FFFFFFF;
  }
}

// Full-list rows, tiled builder (the default): one workgroup per Morton-aligned brick of 4x4x4
// bins.  Beads are sorted by (Morton code of bin, tag), so a brick's beads are one contiguous
// range; every candidate they can see lies in the 8x8x8 bins around the brick.  Those candidates
// (position, index, tag: 32 B) are staged in LDS once, bin by bin (owned, then ghosts), and each
// lane then walks its bead's stencil in the specified (dz, dy, dx) order over LDS, appending the
// kept entries -- tagged with their distance class -- to the bead's column of a scratch buffer
// through one running pointer, and finally copies the column into the row class by class (eight
// loads in flight).  Round 4: the walk used to send class 0 straight to the row and the skin
// classes to a side buffer, with two counters, two capacities and a 64-bit slot address per kept
// entry; one pointer and a byte-packed class count cost half the instructions in the block every
// candidate runs through (1200 -> ~1010 us per re-neighbouring at 1 M beads).  Same rows, bit for
// bit, as k_rows_discover + k_rows_partition (kept as the fallback for stencils wider than 2 bins
// and for bricks whose candidates exceed the staging capacity).
constexpr int TILE_B = 192;      // lanes per brick (mean ~145 beads at rho* = 0.8)
constexpr int TILE_BX = 4;       // brick = 4 x 4 x 4 bins = the low 6 bits of the Morton code (8 x 4 x 4 with
                                 // 320 lanes measured slower: one workgroup per CU instead of three)
constexpr int TILE_RX = TILE_BX + 4, TILE_R = 8;  // region in bins: the brick + 2 either side
constexpr int TILE_NREG = TILE_RX * TILE_R * TILE_R;
constexpr int TILE_PER_LANE = TILE_NREG / 64;     // bins per lane of the scanning wavefront
constexpr int TILE_CAP = 1536;   // staged candidates per brick (mean ~1100 at rho* = 0.8)
static_assert(TILE_NREG % 64 == 0, "region bins are scanned by one wavefront");

struct __attribute__((aligned(16))) TileCand {
  double x, y, z;
  int idx, tag;
};

// One bead's walk over the staged candidates.  For a fixed (dz, dy) the stencil's bins are
// consecutive in x, and so are their staged candidates: 25 contiguous LDS ranges per bead instead
// of 125 bins.  Rows of bins (and their outermost bins in x) that lie farther than cutneigh from
// the bead are skipped; candidates are fetched four at a time and the kept ones -- tagged with their
// build-time distance class in bits 30-31 -- are appended in discovery order to the bead's column of
// a scratch buffer through ONE running pointer (an entry costs a store and a pointer step; the
// class-partitioned row is made from the column afterwards, k_rows_tile).  Returns the number of
// kept entries, which may exceed `cap`: the entries beyond it are counted, not stored.
__device__ __forceinline__ int tile_walk(const DomainDev &D, const double3 &binsize, const int k, const double4 &pk,
                                         const int tk, const int bx, const int by, const int bz, const int r0x,
                                         const int r0y, const int r0z, const int *s_start, const TileCand *s_cand,
                                         int *scratch, const int pitch, const int cap, unsigned &classcount)
{
  unsigned cc = 0;  // entries per class, one byte each (a row that long does not fit anyway)
  const double prune = D.cutneighsq * (1.0 + 1.0e-9) + 1.0e-12;
  const size_t step = (size_t) pitch * sizeof(int);
  char *const base = reinterpret_cast<char *>(scratch + k);
  char *const pend = base + (size_t) cap * step;
  char *p = base;
  const int xlo_bin = max(bx - D.sten[0], 0), xhi_bin = min(bx + D.sten[0], D.nbin[0] - 1);
  for (int dz = -D.sten[2]; dz <= D.sten[2]; dz++) {
    const int cz = bz + dz;
    if (cz < 0 || cz >= D.nbin[2]) continue;
    const double zlo = D.bboxlo[2] + cz * binsize.z;
    const double ez = pk.z < zlo ? zlo - pk.z : (pk.z > zlo + binsize.z ? pk.z - (zlo + binsize.z) : 0.0);
    for (int dy = -D.sten[1]; dy <= D.sten[1]; dy++) {
      const int cy = by + dy;
      if (cy < 0 || cy >= D.nbin[1]) continue;
      const double ylo = D.bboxlo[1] + cy * binsize.y;
      const double ey = pk.y < ylo ? ylo - pk.y : (pk.y > ylo + binsize.y ? pk.y - (ylo + binsize.y) : 0.0);
      const double d2 = ez * ez + ey * ey;
      if (d2 > prune) continue;
      int cx0 = xlo_bin, cx1 = xhi_bin;
      while (cx0 < bx) {  // bins to the left of the bead's own
        const double ex = pk.x - (D.bboxlo[0] + cx0 * binsize.x + binsize.x);
        if (ex > 0.0 && d2 + ex * ex > prune) cx0++;
        else break;
      }
      while (cx1 > bx) {  // and to the right
        const double ex = (D.bboxlo[0] + cx1 * binsize.x) - pk.x;
        if (ex > 0.0 && d2 + ex * ex > prune) cx1--;
        else break;
      }
      const int rrow = ((cz - r0z) * TILE_R + (cy - r0y)) * TILE_RX - r0x;
      const int j0 = s_start[rrow + cx0], j1 = s_start[rrow + cx1 + 1];
      for (int j = j0; j < j1; j += 4) {
        TileCand pm[4];
        double rsq[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          pm[u] = s_cand[min(j + u, j1 - 1)];
          const double delx = pk.x - pm[u].x;
          const double dely = pk.y - pm[u].y;
          const double delz = pk.z - pm[u].z;
          rsq[u] = delx * delx + dely * dely + delz * delz;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool ok = (j + u < j1) && (pm[u].idx != k) && (rsq[u] < D.cutneighsq);
          if (ok) {
            const int in0 = rsq[u] < D.cls_sq[0], in1 = rsq[u] < D.cls_sq[1], in2 = rsq[u] < D.cls_sq[2];
            const int cls = 3 - in0 - in1 - in2;
            const int orient = (tk <= pm[u].tag) ? 1 : 0;
            if (p < pend) *reinterpret_cast<int *>(p) = pm[u].idx | (orient << UCG_ORIENT_BIT) | (cls << 30);
            p += step;
            cc += 1u << (cls << 3);
          }
        }
      }
    }
  }
  classcount = cc;
  return (int) ((size_t) (p - base) / step);
}

__global__ __launch_bounds__(TILE_B) void k_rows_tile(const DomainDev D, const double4 *pos4, const int *tag,
                                                      const int *bin_of, const int4 *cells, int *rowcount, int *neigh,
                                                      int *scratch, int pitch, int cap, const double3 binsize,
                                                      const int3 nbrick, int4 *blockstat, int *fallback)
{
  __shared__ TileCand s_cand[TILE_CAP];
  __shared__ int s_start[TILE_NREG + 1], s_cnt[TILE_NREG];
  __shared__ int s_range[4];  // first bead, end bead, owned beads in the brick's bins, staged candidates
  __shared__ int s_max[TILE_B / 64], s_maxs[TILE_B / 64];
  __shared__ unsigned long long s_tot[TILE_B / 64];
  constexpr int NREG = TILE_NREG;
  const int t = threadIdx.x;
  const int bxb = blockIdx.x % nbrick.x, byb = (blockIdx.x / nbrick.x) % nbrick.y, bzb = blockIdx.x / (nbrick.x * nbrick.y);
  const int r0x = TILE_BX * bxb - 2, r0y = 4 * byb - 2, r0z = 4 * bzb - 2;
  if (t == 0) {
    s_range[0] = 0x7FFFFFFF;
    s_range[1] = 0;
    s_range[2] = 0;
  }
  __syncthreads();
  // the region's bins: candidate counts; the brick's own bins also give its bead range
  for (int r = t; r < NREG; r += TILE_B) {
    const int rx = r % TILE_RX, ry = (r / TILE_RX) % TILE_R, rz = r / (TILE_RX * TILE_R);
    const int cx = r0x + rx, cy = r0y + ry, cz = r0z + rz;
    int n = 0;
    if (cx >= 0 && cx < D.nbin[0] && cy >= 0 && cy < D.nbin[1] && cz >= 0 && cz < D.nbin[2]) {
      const int4 cell = cells[(cz * D.nbin[1] + cy) * D.nbin[0] + cx];
      n = (cell.y - cell.x) + (cell.w - cell.z);
      if (rx >= 2 && rx < 2 + TILE_BX && ry >= 2 && ry < 6 && rz >= 2 && rz < 6 && cell.y > cell.x) {
        atomicMin(&s_range[0], cell.x);
        atomicMax(&s_range[1], cell.y);
        atomicAdd(&s_range[2], cell.y - cell.x);
      }
    }
    s_cnt[r] = n;
  }
  __syncthreads();
  const int kbeg = s_range[0], kend = s_range[1];
  if (s_range[2] == 0) return;  // no owned bead in this brick
  // exclusive scan of the region's counts by the first wavefront
  if (t < 64) {
    int loc[TILE_PER_LANE], sum = 0;
#pragma unroll
    for (int i = 0; i < TILE_PER_LANE; i++) {
      loc[i] = sum;
      sum += s_cnt[t * TILE_PER_LANE + i];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (t >= off) incl += v;
    }
    const int excl = incl - sum;
#pragma unroll
    for (int i = 0; i < TILE_PER_LANE; i++) s_start[t * TILE_PER_LANE + i] = excl + loc[i];
    if (t == 63) s_range[3] = s_start[TILE_NREG] = incl;
  }
  __syncthreads();
  if (s_range[3] > TILE_CAP || s_range[2] != kend - kbeg) {
    if (t == 0) atomicOr(fallback, 1);
    return;
  }
  // stage the candidates bin by bin: owned (ascending tag), then ghosts (ascending tag, shift code)
  for (int r = t; r < NREG; r += TILE_B) {
    if (s_cnt[r] == 0) continue;
    const int rx = r % TILE_RX, ry = (r / TILE_RX) % TILE_R, rz = r / (TILE_RX * TILE_R);
    const int4 cell = cells[((r0z + rz) * D.nbin[1] + (r0y + ry)) * D.nbin[0] + (r0x + rx)];
    int j = s_start[r];
    for (int m = cell.x; m < cell.y; m++, j++) {
      const double4 p = pos4[m];
      TileCand c;
      c.x = p.x; c.y = p.y; c.z = p.z; c.idx = m; c.tag = tag[m];
      s_cand[j] = c;
    }
    for (int m = cell.z; m < cell.w; m++, j++) {
      const double4 p = pos4[m];
      TileCand c;
      c.x = p.x; c.y = p.y; c.z = p.z; c.idx = m; c.tag = tag[m];
      s_cand[j] = c;
    }
  }
  __syncthreads();
  int mx = 0, mxs = 0;
  unsigned long long tot = 0;
  for (int k = kbeg + t; k < kend; k += TILE_B) {
    const double4 pk = pos4[k];
    const int tk = tag[k];
    const int b = bin_of[k];
    const int bx = b % D.nbin[0], by = (b / D.nbin[0]) % D.nbin[1], bz = b / (D.nbin[0] * D.nbin[1]);
    unsigned cc;
    const int cnt = tile_walk(D, binsize, k, pk, tk, bx, by, bz, r0x, r0y, r0z, s_start, s_cand, scratch, pitch, cap, cc);
    if (cnt <= cap) {
      // the row: the column's entries class by class, each class in discovery order (the lane reads back its own
      // writes), eight loads in flight
      const int *col = scratch + k;
      int c0 = (int) (cc & 255u), c1 = (int) ((cc >> 8) & 255u), c2 = (int) ((cc >> 16) & 255u);
      if (cnt >= 256) {  // the byte counters of the walk have wrapped: count the column
        c0 = c1 = c2 = 0;
        for (int e = 0; e < cnt; e++) {
          const int cls = (col[(size_t) e * pitch] >> 30) & 3;
          c0 += cls == 0;
          c1 += cls == 1;
          c2 += cls == 2;
        }
      }
      int p0 = 0, p1 = c0, p2 = c0 + c1, p3 = c0 + c1 + c2;
      for (int e = 0; e < cnt; e += 8) {
        int ent[8];
#pragma unroll
        for (int u = 0; u < 8; u++) ent[u] = col[(size_t) min(e + u, cnt - 1) * pitch];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (e + u < cnt) {
            const int cls = (ent[u] >> 30) & 3;
            const int slot = cls == 0 ? p0 : (cls == 1 ? p1 : (cls == 2 ? p2 : p3));
            p0 += cls == 0;
            p1 += cls == 1;
            p2 += cls == 2;
            p3 += cls == 3;
            neigh[(size_t) slot * pitch + k] = ent[u] & 0x3FFFFFFF;
          }
        }
      }
    }
    rowcount[k] = cnt;
    mx = max(mx, cnt);
    tot += (unsigned long long) cnt;
  }
  for (int off = 32; off > 0; off >>= 1) {
    mx = max(mx, __shfl_down(mx, off, 64));
    mxs = max(mxs, __shfl_down(mxs, off, 64));
    tot += __shfl_down(tot, off, 64);
  }
  if ((t & 63) == 0) {
    s_max[t >> 6] = mx;
    s_maxs[t >> 6] = mxs;
    s_tot[t >> 6] = tot;
  }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < TILE_B / 64; w++) {
      mx = max(mx, s_max[w]);
      mxs = max(mxs, s_maxs[w]);
      tot += s_tot[w];
    }
    // one record per brick (the fold kernel reduces them: same-address atomics from thousands of bricks serialise)
    blockstat[blockIdx.x] = make_int4(mx, mxs, (int) tot, 0);
  }
}

// maxima and total of the per-brick records -> rowstat {max row | total | (fallback flag, untouched) | max skin}
__global__ __launch_bounds__(1024) void k_rowstat_fold(int nblocks, const int4 *blockstat, unsigned long long *rowstat)
{
  __shared__ int s_mx[16], s_ms[16];
  __shared__ unsigned long long s_t[16];
  int mx = 0, ms = 0;
  unsigned long long tot = 0;
  for (int b = threadIdx.x; b < nblocks; b += 1024) {
    const int4 v = blockstat[b];
    mx = max(mx, v.x);
    ms = max(ms, v.y);
    tot += (unsigned long long) v.z;
  }
  for (int off = 32; off > 0; off >>= 1) {
    mx = max(mx, __shfl_down(mx, off, 64));
    ms = max(ms, __shfl_down(ms, off, 64));
    tot += __shfl_down(tot, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_mx[threadIdx.x >> 6] = mx;
    s_ms[threadIdx.x >> 6] = ms;
    s_t[threadIdx.x >> 6] = tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) {
      mx = max(mx, s_mx[w]);
      ms = max(ms, s_ms[w]);
      tot += s_t[w];
    }
    rowstat[0] = (unsigned long long) mx;
    rowstat[1] = tot;
    rowstat[3] = (unsigned long long) ms;
  }
}

__global__ __launch_bounds__(NB) void k_store_xhold(int n, const double4 *pos4, double4 *xhold)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i < n) xhold[i] = pos4[i];
}

__global__ __launch_bounds__(NB) void k_check_distance(const DomainDev D, int n, const double4 *pos4, const double4 *xhold,
                                                      int *blockflags)
{
  // Neighbor::check_distance: any bead moved more than skin/2 since the last build.  One flag per workgroup at
  // its own address (on the steps where thousands of beads cross the threshold, same-address traffic from every
  // wavefront serialises in one L2 channel: 80 us instead of 11); k_flags_any folds them.
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  const int i = blockIdx.x * NB + threadIdx.x;
  bool moved = false;
  if (i < n) {
    const double4 p = pos4[i], h = xhold[i];
    const double delx = p.x - h.x, dely = p.y - h.y, delz = p.z - h.z;
    const double rsq = delx * delx + dely * dely + delz * delz;
    moved = rsq > D.triggersq;
  }
  if (__any(moved) && (threadIdx.x & 63) == 0) s_any = 1;
  __syncthreads();
  if (threadIdx.x == 0) blockflags[blockIdx.x] = s_any;
}

// flag[0] = any workgroup flag set; flag[1] = the pair kernels' sticky table-range flag (read here so that the one small
// download of the re-neighbour decision also tells whether an error poll is due)
__global__ __launch_bounds__(1024) void k_flags_any(int nflags, const int *blockflags, int *flag, const int *pair_err)
{
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  int any = 0;
  for (int b = threadIdx.x; b < nflags; b += 1024) any |= blockflags[b];
  if (__any(any != 0) && (threadIdx.x & 63) == 0) s_any = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    flag[0] = s_any;
    flag[1] = pair_err ? *pair_err : 0;
  }
}

// EXPERIMENT (option "rows_sort_r2", VERDICT round 3 item 5 ii): every row re-ordered by the build-time r^2 of its entries
// -- a refinement of the four distance classes -- so that the lanes of a wavefront, at the same position of their rows, meet
// similar distances and hence neighbouring knots of the tables (fewer LDS bank conflicts of the knot reads).  The row order
// is part of the canonical summation order, so results are NOT the specification's bits: a measurement aid, never the default.
// One lane per bead, insertion sort on its column of the transposed list.
__global__ __launch_bounds__(NB) void k_rows_sort_r2(int n, int pitch, const int *numneigh, int *neigh, const double4 *pos4)
{
  const int k = blockIdx.x * NB + threadIdx.x;
  if (k >= n) return;
  const int cnt = numneigh[k];
  const double4 pk = pos4[k];
  for (int a = 1; a < cnt; a++) {
    const int ea = neigh[(size_t) a * pitch + k];
    const double4 pa = pos4[ea & 0x1FFFFFFF];
    const double ra = (pk.x - pa.x) * (pk.x - pa.x) + (pk.y - pa.y) * (pk.y - pa.y) + (pk.z - pa.z) * (pk.z - pa.z);
    int b = a - 1;
    while (b >= 0) {
      const int eb = neigh[(size_t) b * pitch + k];
      const double4 pb = pos4[eb & 0x1FFFFFFF];
      const double rb = (pk.x - pb.x) * (pk.x - pb.x) + (pk.y - pb.y) * (pk.y - pb.y) + (pk.z - pb.z) * (pk.z - pb.z);
      if (rb <= ra) break;
      neigh[(size_t) (b + 1) * pitch + k] = eb;
      b--;
    }
    neigh[(size_t) (b + 1) * pitch + k] = ea;
  }
}

void setup_bins(Domain &D)
{
  // identical host arithmetic to the specification (oracle/orc_md.c: orc_sim_setup_bins)
  // single-rank rebuild and decomposed borders alike enumerate ONE layer of periodic images (shifts -1, 0, +1)
  for (int d = 0; d < 3; d++)
    if (D.prd[d] < D.cutneigh)
      throw InputError{"periodic box shorter than the ghost cutoff: more than one image layer would be needed"};
  const double target = 0.5 * D.cutneigh;
  D.nbins = 1;
  for (int d = 0; d < 3; d++) {
    D.bboxlo[d] = D.sublo[d] - D.cutneigh;
    D.bboxhi[d] = D.subhi[d] + D.cutneigh;
    const double ext = (D.subhi[d] + D.cutneigh) - D.bboxlo[d];
    int nb = (int) (ext / target);
    if (nb < 1) nb = 1;
    D.nbin[d] = nb;
    D.binsize[d] = ext / nb;
    D.bininv[d] = 1.0 / D.binsize[d];
    int sx = (int) (D.cutneigh * D.bininv[d]);
    if (sx * D.binsize[d] < D.cutneigh) sx++;
    D.sten[d] = sx;
    D.nbins *= nb;
  }
}

void sort_pairs(ucg_ctx *ctx, Domain &D, int n)
{
  size_t bytes = 0;
  UCG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, D.keys_in.get(), D.keys_out.get(), D.vals_in.get(),
                                             D.vals_out.get(), n, 0, 64, ctx->stream));
  D.cub_tmp.reserve(bytes + 16);
  UCG_HIP(hipcub::DeviceRadixSort::SortPairs(D.cub_tmp.get(), bytes, D.keys_in.get(), D.keys_out.get(), D.vals_in.get(),
                                             D.vals_out.get(), n, 0, 64, ctx->stream));
}

template <typename T>
void permute(ucg_ctx *ctx, int n, const int *perm, DevBuf<T> &arr, DevBuf<T> &tmp)
{
  // gather into the scratch buffer, then the two trade places (no copy back).  The scratch buffer is made at least as
  // large as the array it replaces, so the arrays of one family converge to one capacity and nothing is reallocated
  // afterwards; entries beyond n (ghosts) are rebuilt by the caller.
  if (tmp.capacity() < arr.capacity()) tmp.reserve_exact(arr.capacity());  // exact: the family's capacity must not creep
  tmp.reserve((size_t) n);
  hipLaunchKernelGGL(k_gather<T>, dim3(nblk(n)), dim3(NB), 0, ctx->stream, n, perm, arr.get(), tmp.get());
  arr.swap(tmp);
}

// (1)+(3): wrap (optional), key, radix-sort owned beads by (bin, tag), permute the per-bead arrays
void sort_owned(ucg_ctx *ctx, bool wrap)
{
  Domain &D = *ctx->dom;
  hipStream_t st = ctx->stream;
  const int n = ctx->nlocal;
  const DomainDev dd = make_dev(D);
  (void) wrap;
  // (1)+(3) wrap, key, sort owned beads by (bin, tag)
  D.keys_in.reserve((size_t) n);
  D.keys_out.reserve((size_t) n);
  D.vals_in.reserve((size_t) n);
  D.vals_out.reserve((size_t) n);
  hipLaunchKernelGGL(k_wrap_and_key, dim3(nblk(n)), dim3(NB), 0, st, dd, n, ctx->pos4.get(), ctx->tag.get(),
                     D.keys_in.get(), D.vals_in.get());
  sort_pairs(ctx, D, n);
  const int *perm = D.vals_out.get();
  // every per-bead array through the permutation in one launch: gather into scratch buffers, which then trade places with
  // the arrays (no copy back).  A scratch buffer is made at least as large as the array it replaces, so the arrays of one
  // family converge to one capacity and nothing is reallocated afterwards; entries beyond n (ghosts) are rebuilt by the caller.
  auto scratch = [&](auto &arr, auto &tmp) {
    if (tmp.capacity() < arr.capacity()) tmp.reserve_exact(arr.capacity());  // exact: the family's capacity must not creep
    tmp.reserve((size_t) n);
  };
  scratch(ctx->pos4, D.tmp4);
  scratch(ctx->vel4, D.tmp4b);
  scratch(ctx->meta, D.tmpi);
  scratch(ctx->tag, D.tmpi2);
  scratch(ctx->mask, D.tmpi3);
  scratch(ctx->num_ucgstates, D.tmpi4);
  scratch(ctx->ucgml, D.tmpd);
  scratch(ctx->ucgp, D.tmpd2);
  if (ctx->has_mol) scratch(ctx->mol, D.tmpi5);
  D.bin_of.reserve((size_t) n);
  PermuteArgs pa;
  pa.pos_s = ctx->pos4.get(); pa.pos_d = D.tmp4.get();
  pa.vel_s = ctx->vel4.get(); pa.vel_d = D.tmp4b.get();
  pa.meta_s = ctx->meta.get(); pa.meta_d = D.tmpi.get();
  pa.tag_s = ctx->tag.get(); pa.tag_d = D.tmpi2.get();
  pa.mask_s = ctx->mask.get(); pa.mask_d = D.tmpi3.get();
  pa.nst_s = ctx->num_ucgstates.get(); pa.nst_d = D.tmpi4.get();
  pa.ml_s = ctx->ucgml.get(); pa.ml_d = D.tmpd.get();
  pa.up_s = ctx->ucgp.get(); pa.up_d = D.tmpd2.get();
  pa.mol_s = ctx->has_mol ? ctx->mol.get() : nullptr;
  pa.mol_d = ctx->has_mol ? D.tmpi5.get() : nullptr;
  hipLaunchKernelGGL(k_permute_all, dim3(nblk(n)), dim3(NB), 0, st, dd, n, perm, pa, D.bin_of.get());
  UCG_HIP(hipGetLastError());
  ctx->pos4.swap(D.tmp4);
  ctx->vel4.swap(D.tmp4b);
  ctx->meta.swap(D.tmpi);
  ctx->tag.swap(D.tmpi2);
  ctx->mask.swap(D.tmpi3);
  ctx->num_ucgstates.swap(D.tmpi4);
  ctx->ucgml.swap(D.tmpd);
  ctx->ucgp.swap(D.tmpd2);
  if (ctx->has_mol) ctx->mol.swap(D.tmpi5);
}

void build_bins_and_rows(ucg_ctx *ctx);

void rebuild(ucg_ctx *ctx)
{
  Domain &D = *ctx->dom;
  hipStream_t st = ctx->stream;
  const int n = ctx->nlocal;
  if (n <= 0) throw InputError{"ucg_neigh_rebuild: no beads uploaded"};
  setup_bins(D);
  if (D.nbin[0] > 512 || D.nbin[1] > 512 || D.nbin[2] > 512) throw InputError{"more than 512 bins per dimension: too many for the sort key"};
  const DomainDev dd = make_dev(D);

  sort_owned(ctx, true);

  // (4) ghosts: count per workgroup, scan, fill, sort by (bin, tag, code)
  const int nblocks = nblk(n);
  D.counter.reserve((size_t) nblocks + 4);       // per-workgroup image counts
  D.blockoffset.reserve((size_t) nblocks + 4);
  hipLaunchKernelGGL(k_ghost_candidates<false>, dim3(nblocks), dim3(NB), 0, st, dd, n, ctx->pos4.get(), ctx->tag.get(),
                     D.counter.get(), nullptr, nullptr, nullptr, nullptr, nullptr);
  {
    size_t bytes = 0;
    UCG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, D.counter.get(), D.blockoffset.get(), nblocks, st));
    D.cub_tmp.reserve(bytes + 16);
    UCG_HIP(hipcub::DeviceScan::ExclusiveSum(D.cub_tmp.get(), bytes, D.counter.get(), D.blockoffset.get(), nblocks, st));
  }
  int ng = 0;
  {
    int last[2] = {0, 0};
    UCG_HIP(hipMemcpyAsync(&last[0], D.counter.get() + (nblocks - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    UCG_HIP(hipMemcpyAsync(&last[1], D.blockoffset.get() + (nblocks - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    UCG_HIP(hipStreamSynchronize(st));
    ng = last[0] + last[1];
  }
  const size_t nall = (size_t) n + (size_t) ng;
  if (nall >= (size_t) UCG_NEIGHMASK) throw InputError{"too many beads + ghosts for 29-bit neighbour indices"};
  ctx->pos4.reserve(nall, true, st);
  ctx->meta.reserve(nall, true, st);
  ctx->tag.reserve(nall, true, st);
  ctx->ucgp.reserve(nall, true, st);
  D.bin_of.reserve(nall, true, st);
  ctx->ghost_src.reserve((size_t) ng + 1);
  D.ghost_code.reserve((size_t) ng + 1);
  if (ng > 0) {
    D.keys_in.reserve((size_t) ng);
    D.keys_out.reserve((size_t) ng);
    D.vals_in.reserve((size_t) ng);
    D.vals_out.reserve((size_t) ng);
    D.cand_src.reserve((size_t) ng);
    D.cand_code.reserve((size_t) ng);
    hipLaunchKernelGGL(k_ghost_candidates<true>, dim3(nblocks), dim3(NB), 0, st, dd, n, ctx->pos4.get(), ctx->tag.get(),
                       nullptr, D.blockoffset.get(), D.keys_in.get(), D.vals_in.get(), D.cand_src.get(), D.cand_code.get());
    sort_pairs(ctx, D, ng);
    hipLaunchKernelGGL(k_ghost_finalize, dim3(nblk(ng)), dim3(NB), 0, st, ng, n, D.vals_out.get(), D.keys_out.get(),
                       D.cand_src.get(), D.cand_code.get(), ctx->ghost_src.get(), D.ghost_code.get(), D.bin_of.get(),
                       ctx->tag.get(), ctx->meta.get());
    hipLaunchKernelGGL(k_halo_forward, dim3(nblk(ng)), dim3(NB), 0, st, dd, ng, n, ctx->ghost_src.get(), D.ghost_code.get(),
                       ctx->pos4.get(), ctx->meta.get(), ctx->ucgp.get());
    hipLaunchKernelGGL(k_bins_from_pos, dim3(nblk(ng)), dim3(NB), 0, st, dd, ng, n, ctx->pos4.get(), D.bin_of.get());
    if (ctx->has_mol) {  // fix cluster_switch looks at the group bit and the molecule of ghosts too
      ctx->mask.reserve(nall, true, st);
      ctx->mol.reserve(nall, true, st);
      hipLaunchKernelGGL(k_ghost_copy_int2, dim3(nblk(ng)), dim3(NB), 0, st, ng, n, ctx->ghost_src.get(), ctx->mask.get(),
                         ctx->mol.get());
    }
  }
  ctx->nghost = ng;
  ctx->ghost_src_valid = true;

  build_bins_and_rows(ctx);
}

void build_bins_and_rows(ucg_ctx *ctx)
{
  Domain &D = *ctx->dom;
  hipStream_t st = ctx->stream;
  const int n = ctx->nlocal, ng = ctx->nghost;
  DomainDev dd = make_dev(D);
  // (2) bin ranges of both classes
  const size_t nb1 = (size_t) D.nbins + 1;
  D.cells.reserve(nb1);
  UCG_HIP(hipMemsetAsync(D.cells.get(), 0, nb1 * sizeof(int4), st));
  hipLaunchKernelGGL(k_cell_ranges, dim3(nblk(n)), dim3(NB), 0, st, n, 0, D.bin_of.get(), D.cells.get(), 0);
  if (ng > 0) hipLaunchKernelGGL(k_cell_ranges, dim3(nblk(ng)), dim3(NB), 0, st, ng, n, D.bin_of.get(), D.cells.get(), 1);

  // (5) rows.  Row capacity comes from the previous build; a row that does not fit is counted,
  // not stored, and the build repeats with the measured maximum.
  const int pitch = ((n + 63) / 64) * 64;
  ctx->numneigh.reserve((size_t) pitch);
  D.rowstat.reserve(8);
  const double3 bs = make_double3(D.binsize[0], D.binsize[1], D.binsize[2]);
  int cap = D.row_capacity > 0 ? D.row_capacity : 96;
  int maxrow = 0;
  long long total = 0;
  bool tiled = !ctx->rows_untiled && D.sten[0] <= 2 && D.sten[1] <= 2 && D.sten[2] <= 2;
  if (tiled) {
    const int3 nbrick = make_int3((D.nbin[0] + TILE_BX - 1) / TILE_BX, (D.nbin[1] + 3) / 4, (D.nbin[2] + 3) / 4);
    const long long nblocks = (long long) nbrick.x * nbrick.y * nbrick.z;
    for (int attempt = 0; attempt < 4; attempt++) {
      ctx->neigh.reserve((size_t) pitch * (size_t) cap);
      D.scratch.reserve((size_t) pitch * (size_t) cap);
      UCG_HIP(hipMemsetAsync(D.rowstat.get(), 0, 8 * sizeof(unsigned long long), st));
      D.blockstat.reserve((size_t) nblocks + 1);
      UCG_HIP(hipMemsetAsync(D.blockstat.get(), 0, (size_t) nblocks * sizeof(int4), st));
      hipLaunchKernelGGL(k_rows_tile, dim3((unsigned) nblocks), dim3(TILE_B), 0, st, dd, ctx->pos4.get(), ctx->tag.get(),
                         D.bin_of.get(), D.cells.get(), ctx->numneigh.get(), ctx->neigh.get(), D.scratch.get(), pitch, cap,
                         bs, nbrick, D.blockstat.get(), (int *) (D.rowstat.get() + 2));
      hipLaunchKernelGGL(k_rowstat_fold, dim3(1), dim3(1024), 0, st, (int) nblocks, D.blockstat.get(),
                         D.rowstat.get());
      unsigned long long stat[6];
      UCG_HIP(hipMemcpyAsync(stat, D.rowstat.get(), sizeof stat, hipMemcpyDeviceToHost, st));
      UCG_HIP(hipStreamSynchronize(st));
      if (stat[2] & 0xFFFFFFFFull) {  // a brick exceeded the staging capacity: the untiled builder takes over
        tiled = false;
        break;
      }
      maxrow = (int) (stat[0] & 0xFFFFFFFFull);
      total = (long long) stat[1];
      if (maxrow <= cap) break;
      cap = maxrow + 16;
      if (attempt == 3) throw InputError{"neighbour rows keep overflowing their buffers"};
    }
  }
  if (!tiled) {
    D.bpos.reserve((size_t) n + ng);
    hipLaunchKernelGGL(k_builder_records, dim3(nblk((long long) n + ng)), dim3(NB), 0, st, n + ng, ctx->pos4.get(),
                       ctx->tag.get(), D.bpos.get());
    D.rowclass.reserve((size_t) pitch * 3);
    for (int attempt = 0; attempt < 3; attempt++) {
      D.scratch.reserve((size_t) pitch * (size_t) cap);
      UCG_HIP(hipMemsetAsync(D.rowstat.get(), 0, 8 * sizeof(unsigned long long), st));
      hipLaunchKernelGGL(k_rows_discover, dim3(nblk(n)), dim3(NB), 0, st, dd, n, D.bpos.get(), D.bin_of.get(),
                         D.cells.get(), ctx->numneigh.get(), D.rowclass.get(), D.scratch.get(), pitch, cap, bs,
                         (int *) D.rowstat.get(), D.rowstat.get() + 1);
      unsigned long long stat[6];
      UCG_HIP(hipMemcpyAsync(stat, D.rowstat.get(), sizeof stat, hipMemcpyDeviceToHost, st));
      UCG_HIP(hipStreamSynchronize(st));
      maxrow = (int) (stat[0] & 0xFFFFFFFFull);
      total = (long long) stat[1];
      if (maxrow <= cap) break;
      cap = maxrow + 16;  // a row did not fit: grow and rediscover
    }
    if (maxrow <= cap) {
      ctx->neigh.reserve((size_t) pitch * (size_t) (maxrow > 0 ? maxrow : 1));
      hipLaunchKernelGGL(k_rows_partition, dim3(nblk(n)), dim3(NB), 0, st, n, ctx->numneigh.get(), D.rowclass.get(),
                         D.scratch.get(), ctx->neigh.get(), pitch);
    }
  }
  if (maxrow > cap) throw InputError{"neighbour rows keep overflowing their buffers"};
  D.row_capacity = maxrow + 16;
  ctx->list_inum = n;
  ctx->list_pitch = pitch;
  ctx->list_maxrow = maxrow;
  ctx->list_entries = total;
  ctx->list_from_builder = true;
  ctx->list_gen++;

  if (ctx->rows_sort_r2)
    hipLaunchKernelGGL(k_rows_sort_r2, dim3(nblk(n)), dim3(NB), 0, st, n, pitch, ctx->numneigh.get(), ctx->neigh.get(), ctx->pos4.get());
  D.xhold.reserve((size_t) n);
  hipLaunchKernelGGL(k_store_xhold, dim3(nblk(n)), dim3(NB), 0, st, n, ctx->pos4.get(), D.xhold.get());
  UCG_HIP(hipGetLastError());
  // per-bead outputs follow the beads' new order only after the next force evaluation
  ctx->frc4.reserve((size_t) n);
  ctx->scores.reserve((size_t) n);
  D.ago = 0;
  ctx->nrebuild++;
}

void halo_forward(ucg_ctx *ctx)
{
  Domain &D = *ctx->dom;
  if (ctx->nghost <= 0) return;
  const DomainDev dd = make_dev(D);
  hipLaunchKernelGGL(k_halo_forward, dim3(nblk(ctx->nghost)), dim3(NB), 0, ctx->stream, dd, ctx->nghost, ctx->nlocal,
                     ctx->ghost_src.get(), D.ghost_code.get(), ctx->pos4.get(), ctx->meta.get(), ctx->ucgp.get());
  UCG_HIP(hipGetLastError());
}

bool decide(ucg_ctx *ctx)
{
  // upstream Neighbor::decide(): a fix with force_reneighbor (fix cluster_switch) whose
  // next_reneighbor is this step forces the build before `ago` is incremented
  Domain &D = *ctx->dom;
  if (cluster_forces_rebuild(ctx)) return true;
  D.ago++;
  if (D.ago >= D.delay && D.ago % D.every == 0) {
    if (D.check == 0) return true;
    const DomainDev dd = make_dev(D);
    D.blockflags.reserve((size_t) nblk(ctx->nlocal) + 1);
    hipLaunchKernelGGL(k_check_distance, dim3(nblk(ctx->nlocal)), dim3(NB), 0, ctx->stream, dd, ctx->nlocal,
                       ctx->pos4.get(), D.xhold.get(), D.blockflags.get());
    hipLaunchKernelGGL(k_flags_any, dim3(1), dim3(1024), 0, ctx->stream, nblk(ctx->nlocal), D.blockflags.get(),
                       D.counter.get(), (const int *) nullptr);
    int flag = 0;
    UCG_HIP(hipMemcpyAsync(&flag, D.counter.get(), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    UCG_HIP(hipStreamSynchronize(ctx->stream));
    return flag != 0;
  }
  return false;
}


// ------------------------------------------------------------------------------------------
// multi-rank (one process per GPU): bead migration, ghost ("border") construction and the
// per-step forward halo.  Transport is the caller's (torch.distributed all_to_all over RCCL);
// these routines only count, pack and unpack on the device.  The gather formulation needs
// no reverse halo.  Replaces what upstream CommBrick::exchange/borders/forward_comm do with
// the field lists of UCG/atom_vec_ucg.cpp:66-82.

struct AtomRec {  // fields_exchange: everything a bead owns (96 bytes)
  double x, y, z, w, vx, vy, vz, vw, ucgp, ucgml;
  int meta, tag, mask, nstates;  // nstates: bits 0-1 num_ucgstates, bits 2.. atom->molecule when molecule ids are resident
};
struct HaloRec {  // fields_border / fields_comm: x (+shift), ucgl, ucgp, ucgstate, type, tag (48 bytes)
  double x, y, z, w, ucgp;
  int meta, tag;  // meta bits 24..28 carry the shift code at border time
};
static_assert(sizeof(AtomRec) == 96 && sizeof(HaloRec) == 48, "record layout");

__device__ __forceinline__ double proc_bound(const DomainDev &D, int d, int i)
{
  // same expression as the host's sublo/subhi (LAMMPS: boxlo + prd*i/procgrid, last = boxhi)
  return (i >= D.procgrid[d]) ? D.boxhi[d] : D.boxlo[d] + D.prd[d] * i / D.procgrid[d];
}

__device__ __forceinline__ int own_loc(const DomainDev &D, int d, double x)
{
  int loc = 0;
  for (int i = 1; i < D.procgrid[d]; i++)
    if (x >= proc_bound(D, d, i)) loc = i;
  return loc;
}

// owner rank of every bead after the wrap; only LEAVERS (dest != me) take a slot in their
// destination's send block -- stayers never move through the transport
__global__ __launch_bounds__(NB) void k_exchange_dest(const DomainDev D, int n, double4 *pos4, int *counts, int *dest_of,
                                                     int *slot_of)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  double4 p = pos4[i];
  p.x = wrap1(p.x, D.boxlo[0], D.boxhi[0], D.prd[0]);
  p.y = wrap1(p.y, D.boxlo[1], D.boxhi[1], D.prd[1]);
  p.z = wrap1(p.z, D.boxlo[2], D.boxhi[2], D.prd[2]);
  pos4[i] = p;
  const int lx = own_loc(D, 0, p.x), ly = own_loc(D, 1, p.y), lz = own_loc(D, 2, p.z);
  const int r = lx + D.procgrid[0] * (ly + D.procgrid[1] * lz);
  dest_of[i] = r;
  slot_of[i] = (r != D.me) ? atomicAdd(&counts[r], 1) : -1;
}

__global__ __launch_bounds__(NB) void k_exchange_pack(int n, int me, const int *dest_of, const int *slot_of,
                                                     const int *offsets, const double4 *pos4, const double4 *vel4,
                                                     const double *ucgp, const double *ucgml, const int *meta,
                                                     const int *tag, const int *mask, const int *nstates, const int *mol,
                                                     AtomRec *out)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n || dest_of[i] == me) return;
  AtomRec r;
  const double4 p = pos4[i], v = vel4[i];
  r.x = p.x; r.y = p.y; r.z = p.z; r.w = p.w;
  r.vx = v.x; r.vy = v.y; r.vz = v.z; r.vw = v.w;
  r.ucgp = ucgp[i];
  r.ucgml = ucgml[i];
  r.meta = meta[i];
  r.tag = tag[i];
  r.mask = mask[i];
  r.nstates = (nstates[i] & 3) | (mol ? (mol[i] << 2) : 0);
  out[offsets[dest_of[i]] + slot_of[i]] = r;
}

// hole filling: with L leavers, the stayers among the last L beads move into the leavers' slots
// among the first n-L beads (as many of one as of the other); the order is irrelevant, the beads
// are sorted afterwards
__global__ __launch_bounds__(NB) void k_exchange_holes(int n, int nkeep, int me, const int *dest_of, int *cnt, int *holes,
                                                      int *movers)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  const bool leaver = dest_of[i] != me;
  if (i < nkeep && leaver) holes[atomicAdd(&cnt[0], 1)] = i;
  if (i >= nkeep && !leaver) movers[atomicAdd(&cnt[1], 1)] = i;
}

__global__ __launch_bounds__(NB) void k_exchange_move(int nmove, const int *holes, const int *movers, double4 *pos4,
                                                     double4 *vel4, double *ucgp, double *ucgml, int *meta, int *tag,
                                                     int *mask, int *nstates, int *mol)
{
  const int j = blockIdx.x * NB + threadIdx.x;
  if (j >= nmove) return;
  const int dst = holes[j], src = movers[j];
  pos4[dst] = pos4[src];
  vel4[dst] = vel4[src];
  ucgp[dst] = ucgp[src];
  ucgml[dst] = ucgml[src];
  meta[dst] = meta[src];
  tag[dst] = tag[src];
  mask[dst] = mask[src];
  nstates[dst] = nstates[src];
  if (mol) mol[dst] = mol[src];
}

__global__ __launch_bounds__(NB) void k_exchange_unpack(int nrecv, int base, const AtomRec *in, double4 *pos4,
                                                       double4 *vel4, double *ucgp, double *ucgml, int *meta, int *tag,
                                                       int *mask, int *nstates, int *mol)
{
  const int j = blockIdx.x * NB + threadIdx.x;
  if (j >= nrecv) return;
  const AtomRec r = in[j];
  const int i = base + j;
  pos4[i] = make_double4(r.x, r.y, r.z, r.w);
  vel4[i] = make_double4(r.vx, r.vy, r.vz, r.vw);
  ucgp[i] = r.ucgp;
  ucgml[i] = r.ucgml;
  meta[i] = r.meta;
  tag[i] = r.tag;
  mask[i] = r.mask;
  nstates[i] = r.nstates & 3;
  if (mol) mol[i] = r.nstates >> 2;
}

// every (bead, shift) image that falls in some rank's extended sub-box, except the bead itself
// on its own rank.  The (rank, shift) loop is uniform over the wavefront, so slots are handed out
// with ONE atomic per wavefront and destination (ballot + prefix count).  FILL=false counts per
// destination; FILL=true writes the send lists.
template <bool FILL>
__global__ __launch_bounds__(NB) void k_border_candidates(const DomainDev D, double cut, int n, int world,
                                                         const double4 *pos4, int *counts, const int *offsets,
                                                         int *send_src, int *send_code)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  const bool live = i < n;
  double4 p = make_double4(0, 0, 0, 0);
  if (live) p = pos4[i];
  const int lane = threadIdx.x & 63;
  for (int r = 0; r < world; r++) {
    const int lx = r % D.procgrid[0], ly = (r / D.procgrid[0]) % D.procgrid[1], lz = r / (D.procgrid[0] * D.procgrid[1]);
    const double xlo = proc_bound(D, 0, lx) - cut, xhi = proc_bound(D, 0, lx + 1) + cut;
    const double ylo = proc_bound(D, 1, ly) - cut, yhi = proc_bound(D, 1, ly + 1) + cut;
    const double zlo = proc_bound(D, 2, lz) - cut, zhi = proc_bound(D, 2, lz + 1) + cut;
    for (int code = 0; code < 27; code++) {
      if (r == D.me && code == 13) continue;
      const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
      const double xs = p.x + sx * D.prd[0], ys = p.y + sy * D.prd[1], zs = p.z + sz * D.prd[2];
      const bool hit = live && xs >= xlo && xs < xhi && ys >= ylo && ys < yhi && zs >= zlo && zs < zhi;
      const unsigned long long mask = __ballot(hit);
      if (mask == 0ull) continue;
      const int leader = __ffsll((long long) mask) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(&counts[r], __popcll(mask));
      base = __shfl(base, leader, 64);
      if (FILL && hit) {
        const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
        send_src[offsets[r] + slot] = i;
        send_code[offsets[r] + slot] = code;
      }
    }
  }
}

__global__ __launch_bounds__(NB) void k_halo_pack(const DomainDev D, int nsend, const int *send_src, const int *send_code,
                                                 const double4 *pos4, const double *ucgp, const int *meta,
                                                 const int *tag, HaloRec *out, int with_code)
{
  const int j = blockIdx.x * NB + threadIdx.x;
  if (j >= nsend) return;
  const int src = send_src[j], code = send_code[j];
  const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
  const double4 p = pos4[src];
  HaloRec r;
  r.x = p.x + sx * D.prd[0];
  r.y = p.y + sy * D.prd[1];
  r.z = p.z + sz * D.prd[2];
  r.w = p.w;
  r.ucgp = ucgp[src];
  r.meta = (meta[src] & 0xFFFFFF) | (with_code ? (code << 24) : 0);
  r.tag = tag[src];
  out[j] = r;
}

__global__ __launch_bounds__(NB) void k_border_keys(const DomainDev D, int ng, const HaloRec *in, unsigned long long *keys,
                                                   int *vals)
{
  const int j = blockIdx.x * NB + threadIdx.x;
  if (j >= ng) return;
  const HaloRec r = in[j];
  const unsigned long long b = morton_of_bin(D, coord2bin(D, r.x, r.y, r.z));
  const unsigned long long code = (unsigned long long) ((r.meta >> 24) & 31);
  keys[j] = (b << 37) | ((unsigned long long) (unsigned int) r.tag << 5) | code;
  vals[j] = j;
}

__global__ __launch_bounds__(NB) void k_border_finalize(int ng, const int *order, int *perm)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g < ng) perm[g] = order[g];
}

__global__ __launch_bounds__(NB) void k_halo_unpack(int ng, int nlocal, const int *perm, const HaloRec *in, double4 *pos4,
                                                   int *meta, double *ucgp, int *tag, int with_tag)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g >= ng) return;
  const HaloRec r = in[perm[g]];
  pos4[nlocal + g] = make_double4(r.x, r.y, r.z, r.w);
  meta[nlocal + g] = r.meta & 0xFFFFFF;
  ucgp[nlocal + g] = r.ucgp;
  if (with_tag) tag[nlocal + g] = r.tag;
}

// The forward halo of a rank whose own periodic images are among its ghosts (every grid with a dimension of one rank):
// the self block never travels, so it needs neither the send nor the receive buffer.  k_halo_pack_peers packs only the
// records bound for other ranks (send-list positions outside [so_self, so_self + nself)); k_halo_unpack_self makes a ghost
// of the self block straight from its owner bead -- the statements of k_halo_pack followed by those of k_halo_unpack, so
// the same bits -- and takes every other ghost from the receive buffer.  pack + device copy + unpack become one launch
// when the whole halo is the rank's own (one rank), and stay pack / transfer / unpack for the peers' blocks.
__global__ __launch_bounds__(NB) void k_halo_pack_peers(const DomainDev D, int nsend, int so_self, int nself, const int *send_src,
                                                       const int *send_code, const double4 *pos4, const double *ucgp,
                                                       const int *meta, const int *tag, HaloRec *out)
{
  int j = blockIdx.x * NB + threadIdx.x;
  if (j >= nsend - nself) return;
  if (j >= so_self) j += nself;
  const int src = send_src[j], code = send_code[j];
  const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
  const double4 p = pos4[src];
  HaloRec r;
  r.x = p.x + sx * D.prd[0];
  r.y = p.y + sy * D.prd[1];
  r.z = p.z + sz * D.prd[2];
  r.w = p.w;
  r.ucgp = ucgp[src];
  r.meta = meta[src] & 0xFFFFFF;
  r.tag = tag[src];
  out[j] = r;
}

__global__ __launch_bounds__(NB) void k_halo_unpack_self(const DomainDev D, int ng, int nlocal, const int *perm, int ro_self,
                                                        int so_self, int nself, const int *send_src, const int *send_code,
                                                        const HaloRec *in, double4 *pos4, int *meta, double *ucgp)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g >= ng) return;
  const int j = perm[g];
  const unsigned js = (unsigned) (j - ro_self);
  if (js < (unsigned) nself) {
    const int src = send_src[so_self + (int) js], code = send_code[so_self + (int) js];
    const int sx = code % 3 - 1, sy = (code / 3) % 3 - 1, sz = code / 9 - 1;
    const double4 p = pos4[src];
    pos4[nlocal + g] = make_double4(p.x + sx * D.prd[0], p.y + sy * D.prd[1], p.z + sz * D.prd[2], p.w);
    meta[nlocal + g] = meta[src] & 0xFFFFFF;
    ucgp[nlocal + g] = ucgp[src];
  } else {
    const HaloRec r = in[j];
    pos4[nlocal + g] = make_double4(r.x, r.y, r.z, r.w);
    meta[nlocal + g] = r.meta & 0xFFFFFF;
    ucgp[nlocal + g] = r.ucgp;
  }
}

// auxiliary forward halo of one double2 per bead (the density style's priors / CV forces)
__global__ __launch_bounds__(NB) void k_halo_aux_pack(int nsend, const int *send_src, const double2 *src, double2 *out)
{
  const int j = blockIdx.x * NB + threadIdx.x;
  if (j < nsend) out[j] = src[send_src[j]];
}

__global__ __launch_bounds__(NB) void k_halo_aux_unpack(int ng, int nlocal, const int *perm, const double2 *in, double2 *dst)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g < ng) dst[nlocal + g] = in[perm[g]];
}

__global__ __launch_bounds__(NB) void k_halo_molmask_pack(int nsend, const int *send_src, const int *mask, const int *mol,
                                                         int2 *out)
{
  const int j = blockIdx.x * NB + threadIdx.x;
  if (j < nsend) out[j] = make_int2(mask[send_src[j]], mol[send_src[j]]);
}

__global__ __launch_bounds__(NB) void k_halo_molmask_unpack(int ng, int nlocal, const int *perm, const int2 *in, int *mask,
                                                           int *mol)
{
  const int g = blockIdx.x * NB + threadIdx.x;
  if (g >= ng) return;
  const int2 v = in[perm[g]];
  mask[nlocal + g] = v.x;
  mol[nlocal + g] = v.y;
}

void counts_to_host(ucg_ctx *ctx, Domain &D, long long *out)
{
  std::vector<int> h((size_t) D.world);
  UCG_HIP(hipMemcpyAsync(h.data(), D.counter.get(), (size_t) D.world * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
  for (int r = 0; r < D.world; r++) out[r] = h[(size_t) r];
}

void offsets_to_device(ucg_ctx *ctx, Domain &D, const std::vector<long long> &counts, DevBuf<int> &dst)
{
  if (D.h_off_cap < D.world + 1) {
    for (int k = 0; k < 2; k++) {
      if (D.h_off[k]) (void) hipHostFree(D.h_off[k]);
      D.h_off[k] = nullptr;
      UCG_HIP(hipHostMalloc((void **) &D.h_off[k], (size_t) (D.world + 1) * sizeof(int), hipHostMallocDefault));
    }
    D.h_off_cap = D.world + 1;
  }
  int *off = D.h_off[D.h_off_next];
  D.h_off_next ^= 1;
  off[0] = 0;
  for (int r = 0; r < D.world; r++) off[r + 1] = off[r] + (int) counts[(size_t) r];
  dst.reserve((size_t) D.world + 1);
  UCG_HIP(hipMemcpyAsync(dst.get(), off, (size_t) (D.world + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
}

template <typename F>
int guarded(ucg_ctx *ctx, F &&fn)
{
  try {
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

int need_domain(ucg_ctx *ctx)
{
  if (!ctx->dom) {
    ctx->err = "ucg_domain_set has not been called";
    return UCG_ERR_INVALID;
  }
  return UCG_OK;
}

// one force evaluation + post_force fixes, in fix-definition order (thermostat before ucgstate,
// UCG/fix_ucgstate.cpp:143-154)
int forces_and_post_force(ucg_ctx *ctx, int ev)
{
  int rc = ucg_pair_compute(ctx->md_pair, ev, ev, nullptr, nullptr);
  if (rc) return rc;
  if (ctx->md_lang) {
    rc = ucg_fix_langevin_post_force(ctx, ctx->groupbit, ctx->ntimestep, ctx->beginstep, ctx->endstep);
    if (rc) return rc;
  }
  if (ctx->md_ucgst) {
    rc = ucg_fix_ucgstate_post_force(ctx);
    if (rc) return rc;
  }
  return UCG_OK;
}

}  // namespace

// per-step forward halo with the rank's self block kept off the buffers (csrc/ucg_comm.hip: multi_halo_forward)
int halo_pack_peers(ucg_ctx *ctx, void *sendbuf, long long so_self, long long nself)
{
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const long long npeer = D.nsend - nself;
    if (npeer <= 0) return UCG_OK;
    if (!sendbuf) return UCG_ERR_INVALID;
    const DomainDev dd = make_dev(D);
    hipLaunchKernelGGL(k_halo_pack_peers, dim3(nblk(npeer)), dim3(NB), 0, ctx->stream, dd, (int) D.nsend, (int) so_self, (int) nself,
                       D.send_src.get(), D.send_code.get(), ctx->pos4.get(), ctx->ucgp.get(), ctx->meta.get(), ctx->tag.get(),
                       (HaloRec *) sendbuf);
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int halo_unpack_self(ucg_ctx *ctx, const void *recvbuf, long long ro_self, long long so_self, long long nself)
{
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const int ng = ctx->nghost;
    if (ng == 0) return UCG_OK;
    if (!recvbuf && nself < ng) return UCG_ERR_INVALID;
    const DomainDev dd = make_dev(D);
    hipLaunchKernelGGL(k_halo_unpack_self, dim3(nblk(ng)), dim3(NB), 0, ctx->stream, dd, ng, ctx->nlocal, D.ghost_perm.get(),
                       (int) ro_self, (int) so_self, (int) nself, D.send_src.get(), D.send_code.get(), (const HaloRec *) recvbuf,
                       ctx->pos4.get(), ctx->meta.get(), ctx->ucgp.get());
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

}  // namespace ucg

using namespace ucg;

extern "C" {

int ucg_domain_set(ucg_ctx *ctx, const double *boxlo, const double *boxhi, double cutforce, double skin, int every,
                   int delay, int check)
{
  if (!ctx || !boxlo || !boxhi) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (!(cutforce > 0.0) || skin < 0.0 || every < 1 || delay < 0) throw InputError{"Illegal neighbor / neigh_modify settings"};
    if (!ctx->dom) ctx->dom = new Domain();
    Domain &D = *ctx->dom;
    for (int d = 0; d < 3; d++) {
      if (!(boxhi[d] > boxlo[d])) throw InputError{"Box bounds are invalid"};
      D.boxlo[d] = boxlo[d];
      D.boxhi[d] = boxhi[d];
      D.prd[d] = boxhi[d] - boxlo[d];
    }
    for (int d = 0; d < 3; d++) {
      D.procgrid[d] = 1;
      D.myloc[d] = 0;
      D.sublo[d] = D.boxlo[d];
      D.subhi[d] = D.boxhi[d];
    }
    D.me = 0;
    D.world = 1;
    ctx->dom_world = 1;
    D.cutforce = cutforce;
    D.skin = skin;
    ctx->skin = skin;
    D.cutneigh = cutforce + skin;
    D.every = every;
    D.delay = delay;
    D.check = check;
    D.counter.reserve(4);
    return UCG_OK;
  });
}

int ucg_neigh_rebuild(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    mirror_need(ctx, UCG_F_ALL);  // whatever the caller changed takes part in the re-ordering
    rebuild(ctx);
    mirror_wrote(ctx, UCG_F_ALL);  // the beads have a new order: every host mirror is behind
    return UCG_OK;
  });
}

int ucg_halo_forward(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    // fields_comm (UCG/atom_vec_ucg.cpp:71): the ghosts get their owners' x, ucgstate, ucgl, ucgp -- a host-side edit of any
    // of them announced with ucg_host_modified has to be on the device first, or owners and ghosts would disagree
    mirror_need(ctx, UCG_F_X | UCG_F_STATE | UCG_F_UCGL | UCG_F_UCGP);
    halo_forward(ctx);
    return UCG_OK;
  });
}

int ucg_ghosts_download(ucg_ctx *ctx, int *src, int *shift3, int cap)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    const int ng = ctx->nghost;
    if (cap < ng) {
      ctx->err = "ghost buffers too small";
      return UCG_ERR_INVALID;
    }
    std::vector<int> s((size_t) ng), c((size_t) ng);
    if (ng) {
      UCG_HIP(hipMemcpyAsync(s.data(), ctx->ghost_src.get(), (size_t) ng * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      UCG_HIP(hipMemcpyAsync(c.data(), ctx->dom->ghost_code.get(), (size_t) ng * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      UCG_HIP(hipStreamSynchronize(ctx->stream));
    }
    for (int g = 0; g < ng; g++) {
      if (src) src[g] = s[(size_t) g];
      if (shift3) {
        const int code = c[(size_t) g];
        shift3[3 * g + 0] = code % 3 - 1;
        shift3[3 * g + 1] = (code / 3) % 3 - 1;
        shift3[3 * g + 2] = code / 9 - 1;
      }
    }
    return UCG_OK;
  });
}

/* host-built ghosts of a single rank as periodic images (the drop-in path): source owned atom + box shifts per ghost.
 * From here on the device can refresh the ghosts (ucg_halo_forward: x + shift * prd like CommBrick's pbc flags) and take
 * the re-neighbour decision (ucg_decide_local) against the positions held now, as after a device build. */
int ucg_ghosts_upload_images(ucg_ctx *ctx, const int *src, const int *shift3, int nghost)
{
  if (!ctx || nghost < 0 || (nghost > 0 && (!src || !shift3))) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    if (D.world != 1) throw InputError{"ucg_ghosts_upload_images: single-rank runs (ghosts of a decomposed run belong to other ranks)"};
    if (nghost != ctx->nghost) throw InputError{"ghost map length differs from the resident ghost count"};
    std::vector<int> code((size_t) nghost + 1, 13);
    for (int g = 0; g < nghost; g++) {
      if (src[g] < 0 || src[g] >= ctx->nlocal) throw InputError{"ghost source index out of range"};
      int c = 0, mul = 1;
      for (int d = 0; d < 3; d++) {
        const int sh = shift3[3 * g + d];
        if (sh < -1 || sh > 1) throw InputError{"ghost image shift outside -1 .. 1"};
        c += (sh + 1) * mul;
        mul *= 3;
      }
      code[(size_t) g] = c;
    }
    ctx->ghost_src.reserve((size_t) nghost + 1);
    D.ghost_code.reserve((size_t) nghost + 1);
    if (nghost) {
      UCG_HIP(hipMemcpyAsync(ctx->ghost_src.get(), src, (size_t) nghost * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
      UCG_HIP(hipMemcpyAsync(D.ghost_code.get(), code.data(), (size_t) nghost * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    D.xhold.reserve((size_t) ctx->nlocal + 1);
    if (ctx->nlocal)
      hipLaunchKernelGGL(k_store_xhold, dim3(nblk(ctx->nlocal)), dim3(NB), 0, ctx->stream, ctx->nlocal, ctx->pos4.get(), D.xhold.get());
    UCG_HIP(hipGetLastError());
    UCG_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ghost_src_valid = true;
    D.ago = 0;
    return UCG_OK;
  });
}

int ucg_md_attach(ucg_ctx *ctx, ucg_pair *pair, int use_nve, int use_langevin, int use_ucgstate)
{
  if (!ctx || !pair || pair->ctx != ctx) return UCG_ERR_INVALID;
  ctx->md_pair = pair;
  if (use_nve < 0 || use_nve > 2) return UCG_ERR_INVALID;
  ctx->md_nve = use_nve;  // 1: fix nve/ucgld, 2: fix nve/ucgld/wall/hard
  ctx->md_lang = use_langevin != 0;
  ctx->md_ucgst = use_ucgstate != 0;
  if (ctx->md_lang && !ctx->lang.active) {
    ctx->err = "ucg_md_attach: fix ucgld/langevin requested but not created";
    return UCG_ERR_INVALID;
  }
  if (ctx->md_ucgst && !ctx->ucgst.active) {
    ctx->err = "ucg_md_attach: fix ucgstate requested but not created";
    return UCG_ERR_INVALID;
  }
  return UCG_OK;
}

int ucg_md_setup(ucg_ctx *ctx, long long nsteps_planned)
{
  if (!ctx || !ctx->md_pair) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  // Verlet::setup(): lists, forces, then each fix's setup(); Fix_UCGLD_Langevin::setup and
  // FixUCGState::setup both call post_force (UCG/fix_ucgld_langevin.cpp:187-197, UCG/fix_ucgstate.cpp:142-171)
  ctx->beginstep = ctx->ntimestep;
  ctx->endstep = ctx->ntimestep + nsteps_planned;
  // resident driver: bound host mirrors are read once and fall behind as a whole
  if (int rc = guarded(ctx, [&]() -> int { mirror_need(ctx, UCG_F_ALL); mirror_wrote(ctx, UCG_F_ALL); return UCG_OK; })) return rc;
  if (ctx->comm) return md_setup_multi(ctx);  // decomposed run: csrc/ucg_comm.hip
  int rc = guarded(ctx, [&]() -> int {
    if (ctx->md_lang && !ctx->lang.inited) {
      // Fix_UCGLD_Langevin::init(): reads atom->ucgml[1..ntypes] of the CURRENT bead order (App. B #5)
      std::vector<double> ml((size_t) ctx->ntypes + 1, 0.0);
      const int cnt = ctx->ntypes + 1 <= ctx->nlocal ? ctx->ntypes + 1 : ctx->nlocal;
      UCG_HIP(hipMemcpyAsync(ml.data(), ctx->ucgml.get(), (size_t) cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      UCG_HIP(hipStreamSynchronize(ctx->stream));
      if (int r = ucg_fix_langevin_init_from_ucgml(ctx, ctx->ntypes, ml.data())) return r;
    }
    rebuild(ctx);
    return UCG_OK;
  });
  if (rc) return rc;
  if ((rc = forces_and_post_force(ctx, 1))) return rc;
  return ucg_pair_check_errors(ctx->md_pair);
}

static int md_run_impl(ucg_ctx *ctx, long long nsteps, int thermo_every, int ev_on_last);

int ucg_md_run(ucg_ctx *ctx, long long nsteps, int thermo_every) { return md_run_impl(ctx, nsteps, thermo_every, 0); }

int ucg_md_run_until(ucg_ctx *ctx, long long nsteps, int ev_on_last) { return md_run_impl(ctx, nsteps, 0, ev_on_last ? 1 : 0); }

int ucg_md_set_window(ucg_ctx *ctx, long long beginstep, long long endstep)
{
  if (!ctx || endstep < beginstep) return UCG_ERR_INVALID;
  ctx->beginstep = beginstep;
  ctx->endstep = endstep;
  return UCG_OK;
}

static int md_run_impl(ucg_ctx *ctx, long long nsteps, int thermo_every, int ev_on_last)
{
  if (!ctx || !ctx->md_pair) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  if (int rc = guarded(ctx, [&]() -> int { mirror_need(ctx, UCG_F_ALL); mirror_wrote(ctx, UCG_F_ALL); return UCG_OK; })) return rc;
  if (ctx->comm) return md_run_multi(ctx, nsteps, thermo_every, ev_on_last);  // decomposed run: csrc/ucg_comm.hip
  // Per step (upstream Verlet::run):
  //   initial_integrate | decide -> rebuild or halo refresh | pair | post_force fixes | final_integrate | end_of_step
  // The per-bead hooks after the pair kernel run as ONE fused kernel; when no thermo output has to
  // look at the state between two steps, that kernel also performs the next step's
  // initial_integrate (same forces, same statement order: bit-identical, one HBM pass fewer).
  bool initial_done = false;
  for (long long s = 0; s < nsteps; s++) {
    ctx->ntimestep++;
    const int ev = ((thermo_every > 0 && (ctx->ntimestep % thermo_every == 0)) || (ev_on_last && s + 1 == nsteps)) ? 1 : 0;
    int rc;
    if (ctx->md_nve && !initial_done &&
        (rc = ctx->md_nve == 2 ? ucg_fix_nve_wall_hard_initial(ctx, ctx->groupbit) : ucg_fix_nve_initial(ctx, ctx->groupbit)))
      return rc;
    initial_done = false;
    rc = guarded(ctx, [&]() -> int {
      if (decide(ctx)) {
        // a table-range violation (the reference's error->one, UCG/pair_table_ucgld.cpp:436-444) must not go
        // unnoticed while the trajectory runs on: the sticky flag is read where the stream is drained anyway
        if (int e = ucg_pair_check_errors(ctx->md_pair)) return e;
        rebuild(ctx);
        // FixClusterSwitch::pre_exchange (UCG/fix_cluster_switch.cpp:452-469): its own exchange /
        // borders / build and Verlet's see the same positions, so one rebuild serves both
        if (cluster_forces_rebuild(ctx)) {
          cluster_pre_exchange(ctx);
          halo_forward(ctx);  // comm->forward_comm(this): the ghosts' new types
        }
      } else
        halo_forward(ctx);
      return UCG_OK;
    });
    if (rc) return rc;
    const bool fuse_next = ctx->md_nve && !ev && (s + 1 < nsteps) && !ctx->md_no_fuse;
    // a step whose next initial_integrate is fused in runs its per-bead hooks in the gather kernel's epilogue
    rc = fuse_next ? ucg_md_pair_post(ctx, ctx->md_pair, ctx->md_lang, ctx->md_ucgst, ctx->md_nve, ctx->groupbit,
                                      ctx->ntimestep, ctx->beginstep, ctx->endstep)
                   : UCG_ERR_UNSUPPORTED;
    if (rc == UCG_ERR_UNSUPPORTED) {
      if ((rc = ucg_pair_compute(ctx->md_pair, ev, ev, nullptr, nullptr))) return rc;
      rc = ucg_md_post_fused(ctx, ctx->md_lang, ctx->md_ucgst, ctx->md_nve, fuse_next, ctx->groupbit, ctx->ntimestep,
                             ctx->beginstep, ctx->endstep);
    }
    if (rc) return rc;
    initial_done = fuse_next;
    // end_of_step: the lambda temperature is a diagnostic (compute_scalar); evaluated on thermo steps
    if (ev && ctx->md_lang && (rc = ucg_fix_langevin_end_of_step(ctx, ctx->groupbit, nullptr))) return rc;
    if (ev && (rc = ucg_pair_check_errors(ctx->md_pair))) return rc;
  }
  return ucg_pair_check_errors(ctx->md_pair);
}

int ucg_md_info(ucg_ctx *ctx, long long *out)
{
  if (!ctx || !out) return UCG_ERR_INVALID;
  for (int i = 0; i < 16; i++) out[i] = 0;
  out[0] = ctx->ntimestep;
  out[1] = ctx->nrebuild;
  out[2] = ctx->nlocal;
  out[3] = ctx->nghost;
  out[4] = ctx->list_entries;
  out[5] = ctx->pair_error_steps;
  out[6] = ctx->list_maxrow;
  out[7] = ctx->list_pitch;
  if (ctx->dom) {
    out[8] = ctx->dom->nbin[0];
    out[9] = ctx->dom->nbin[1];
    out[10] = ctx->dom->nbin[2];
  }
  return UCG_OK;
}

int ucg_md_thermo(ucg_ctx *ctx, double *out9)
{
  if (!ctx || !out9) return UCG_ERR_INVALID;
  for (int i = 0; i < 9; i++) out9[i] = ctx->thermo[i];
  return UCG_OK;
}

/* ----------------------------------------------------------------- multi-rank support */

int ucg_decomp_set(ucg_ctx *ctx, const int *procgrid, int me)
{
  if (!ctx || !procgrid) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const int world = procgrid[0] * procgrid[1] * procgrid[2];
    if (procgrid[0] < 1 || procgrid[1] < 1 || procgrid[2] < 1 || me < 0 || me >= world)
      throw InputError{"Bad processor grid"};
    D.world = world;
    ctx->dom_world = world;
    D.me = me;
    for (int d = 0; d < 3; d++) D.procgrid[d] = procgrid[d];
    D.myloc[0] = me % procgrid[0];
    D.myloc[1] = (me / procgrid[0]) % procgrid[1];
    D.myloc[2] = me / (procgrid[0] * procgrid[1]);
    for (int d = 0; d < 3; d++) {
      D.sublo[d] = D.boxlo[d] + D.prd[d] * D.myloc[d] / D.procgrid[d];
      D.subhi[d] = (D.myloc[d] + 1 >= D.procgrid[d]) ? D.boxhi[d] : D.boxlo[d] + D.prd[d] * (D.myloc[d] + 1) / D.procgrid[d];
      if (D.subhi[d] - D.sublo[d] < D.cutneigh)
        throw InputError{"sub-domain thinner than the ghost cutoff: one halo layer would not be enough"};
    }
    D.counter.reserve((size_t) world + 4);
    D.send_counts.assign((size_t) world, 0);
    D.recv_counts.assign((size_t) world, 0);
    return UCG_OK;
  });
}

}  // extern "C"

namespace ucg {
// The two halves of ucg_exchange_count / ucg_border_count for a communicator that exchanges the counts ON THE DEVICE
// (csrc/ucg_comm.hip, RCCL transport: the per-destination counters go from the kernel that made them straight into the
// all-to-all, and the host reads its send and receive counts with one download): *_launch queues the counting kernels
// and returns the device counters (world ints), counts_adopt installs the host copy afterwards.
int exchange_count_launch(ucg_ctx *ctx, const int **dev_counts)
{
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const int n = ctx->nlocal;
    const DomainDev dd = make_dev(D);
    D.counter.reserve((size_t) D.world + 4);
    D.dest_of.reserve((size_t) n + 1);
    D.slot_of.reserve((size_t) n + 1);
    UCG_HIP(hipMemsetAsync(D.counter.get(), 0, (size_t) D.world * sizeof(int), ctx->stream));
    if (n > 0)
      hipLaunchKernelGGL(k_exchange_dest, dim3(nblk(n)), dim3(NB), 0, ctx->stream, dd, n, ctx->pos4.get(),
                         D.counter.get(), D.dest_of.get(), D.slot_of.get());
    UCG_HIP(hipGetLastError());
    if (dev_counts) *dev_counts = D.counter.get();
    return UCG_OK;
  });
}

int border_count_launch(ucg_ctx *ctx, const int **dev_counts)
{
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const int n = ctx->nlocal;
    setup_bins(D);
    if (D.nbin[0] > 512 || D.nbin[1] > 512 || D.nbin[2] > 512) throw InputError{"more than 512 bins per dimension: too many for the sort key"};
    if (n > 0) sort_owned(ctx, true);
    const DomainDev dd = make_dev(D);
    UCG_HIP(hipMemsetAsync(D.counter.get(), 0, (size_t) D.world * sizeof(int), ctx->stream));
    if (n > 0)
      hipLaunchKernelGGL(k_border_candidates<false>, dim3(nblk(n)), dim3(NB), 0, ctx->stream, dd, D.cutneigh, n, D.world,
                         ctx->pos4.get(), D.counter.get(), nullptr, nullptr, nullptr);
    UCG_HIP(hipGetLastError());
    if (dev_counts) *dev_counts = D.counter.get();
    return UCG_OK;
  });
}

// which: 0 = the leavers' counts of exchange_count_launch, 1 = the ghosts' counts of border_count_launch
int counts_adopt(ucg_ctx *ctx, int which, const long long *sendcounts)
{
  if (int rc = need_domain(ctx)) return rc;
  Domain &D = *ctx->dom;
  if (which == 1) D.nsend = 0;
  for (int r = 0; r < D.world; r++) {
    D.send_counts[(size_t) r] = sendcounts[r];
    if (which == 1) D.nsend += sendcounts[r];
  }
  return UCG_OK;
}
}  // namespace ucg

extern "C" {

int ucg_exchange_count(ucg_ctx *ctx, long long *sendcounts)
{
  if (!ctx || !sendcounts) return UCG_ERR_INVALID;
  if (int rc = exchange_count_launch(ctx, nullptr)) return rc;
  return guarded(ctx, [&]() -> int {
    counts_to_host(ctx, *ctx->dom, sendcounts);  // leavers only: sendcounts[me] == 0
    return counts_adopt(ctx, 0, sendcounts);
  });
}

int ucg_exchange_pack(ucg_ctx *ctx, void *sendbuf)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    hipStream_t st = ctx->stream;
    const int n = ctx->nlocal;
    long long nleave = 0;
    for (int r = 0; r < D.world; r++) nleave += D.send_counts[(size_t) r];
    if (nleave == 0) return UCG_OK;
    if (!sendbuf) return UCG_ERR_INVALID;
    offsets_to_device(ctx, D, D.send_counts, D.offsets);
    hipLaunchKernelGGL(k_exchange_pack, dim3(nblk(n)), dim3(NB), 0, st, n, D.me, D.dest_of.get(), D.slot_of.get(),
                       D.offsets.get(), ctx->pos4.get(), ctx->vel4.get(), ctx->ucgp.get(), ctx->ucgml.get(),
                       ctx->meta.get(), ctx->tag.get(), ctx->mask.get(), ctx->num_ucgstates.get(),
                       ctx->has_mol ? ctx->mol.get() : nullptr, (AtomRec *) sendbuf);
    // close the holes the leavers leave behind
    const int nkeep = n - (int) nleave;
    D.holes.reserve((size_t) nleave + 1);
    D.movers.reserve((size_t) nleave + 1);
    UCG_HIP(hipMemsetAsync(D.counter.get(), 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(k_exchange_holes, dim3(nblk(n)), dim3(NB), 0, st, n, nkeep, D.me, D.dest_of.get(), D.counter.get(),
                       D.holes.get(), D.movers.get());
    // both lists have the same length (<= nleave); a launch over nleave lanes with a device-side bound
    int cnt[2] = {0, 0};
    UCG_HIP(hipMemcpyAsync(cnt, D.counter.get(), sizeof cnt, hipMemcpyDeviceToHost, st));
    UCG_HIP(hipStreamSynchronize(st));
    if (cnt[0] != cnt[1]) throw InputError{"internal error: bead migration holes and movers differ"};
    if (cnt[0] > 0)
      hipLaunchKernelGGL(k_exchange_move, dim3(nblk(cnt[0])), dim3(NB), 0, st, cnt[0], D.holes.get(), D.movers.get(),
                         ctx->pos4.get(), ctx->vel4.get(), ctx->ucgp.get(), ctx->ucgml.get(), ctx->meta.get(),
                         ctx->tag.get(), ctx->mask.get(), ctx->num_ucgstates.get(), ctx->has_mol ? ctx->mol.get() : nullptr);
    UCG_HIP(hipGetLastError());
    ctx->nlocal = nkeep;
    return UCG_OK;
  });
}

int ucg_exchange_unpack(ucg_ctx *ctx, const void *recvbuf, long long nrecv)
{
  if (!ctx || nrecv < 0 || (nrecv > 0 && !recvbuf)) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    hipStream_t st = ctx->stream;
    const int base = ctx->nlocal;
    const size_t n = (size_t) base + (size_t) nrecv;
    ctx->pos4.reserve(n, true, st);
    ctx->vel4.reserve(n, true, st);
    ctx->frc4.reserve(n);
    ctx->scores.reserve(n);
    ctx->ucgp.reserve(n, true, st);
    ctx->ucgml.reserve(n, true, st);
    ctx->meta.reserve(n, true, st);
    ctx->tag.reserve(n, true, st);
    ctx->mask.reserve(n, true, st);
    if (ctx->has_mol) ctx->mol.reserve(n, true, st);
    ctx->num_ucgstates.reserve(n, true, st);
    if (nrecv)
      hipLaunchKernelGGL(k_exchange_unpack, dim3(nblk(nrecv)), dim3(NB), 0, st, (int) nrecv, base,
                         (const AtomRec *) recvbuf, ctx->pos4.get(), ctx->vel4.get(), ctx->ucgp.get(), ctx->ucgml.get(),
                         ctx->meta.get(), ctx->tag.get(), ctx->mask.get(), ctx->num_ucgstates.get(),
                         ctx->has_mol ? ctx->mol.get() : nullptr);
    UCG_HIP(hipGetLastError());
    ctx->nlocal = (int) n;
    ctx->nghost = 0;
    ctx->list_inum = 0;
    return UCG_OK;
  });
}

int ucg_border_count(ucg_ctx *ctx, long long *sendcounts)
{
  if (!ctx || !sendcounts) return UCG_ERR_INVALID;
  if (int rc = border_count_launch(ctx, nullptr)) return rc;
  return guarded(ctx, [&]() -> int {
    counts_to_host(ctx, *ctx->dom, sendcounts);
    return counts_adopt(ctx, 1, sendcounts);
  });
}

int ucg_border_pack(ucg_ctx *ctx, void *sendbuf)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const int n = ctx->nlocal;
    if (D.nsend == 0) return UCG_OK;
    if (!sendbuf) return UCG_ERR_INVALID;
    const DomainDev dd = make_dev(D);
    offsets_to_device(ctx, D, D.send_counts, D.offsets);
    D.send_src.reserve((size_t) D.nsend);
    D.send_code.reserve((size_t) D.nsend);
    UCG_HIP(hipMemsetAsync(D.counter.get(), 0, (size_t) D.world * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_border_candidates<true>, dim3(nblk(n)), dim3(NB), 0, ctx->stream, dd, D.cutneigh, n, D.world,
                       ctx->pos4.get(), D.counter.get(), D.offsets.get(), D.send_src.get(), D.send_code.get());
    hipLaunchKernelGGL(k_halo_pack, dim3(nblk(D.nsend)), dim3(NB), 0, ctx->stream, dd, (int) D.nsend, D.send_src.get(),
                       D.send_code.get(), ctx->pos4.get(), ctx->ucgp.get(), ctx->meta.get(), ctx->tag.get(),
                       (HaloRec *) sendbuf, 1);
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int ucg_border_unpack(ucg_ctx *ctx, const void *recvbuf, long long nrecv)
{
  if (!ctx || nrecv < 0 || (nrecv > 0 && !recvbuf)) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    hipStream_t st = ctx->stream;
    const int n = ctx->nlocal, ng = (int) nrecv;
    const size_t nall = (size_t) n + (size_t) ng;
    if (nall >= (size_t) UCG_NEIGHMASK) throw InputError{"too many beads + ghosts for 29-bit neighbour indices"};
    const DomainDev dd = make_dev(D);
    ctx->pos4.reserve(nall, true, st);
    ctx->meta.reserve(nall, true, st);
    ctx->tag.reserve(nall, true, st);
    ctx->ucgp.reserve(nall, true, st);
    D.bin_of.reserve(nall, true, st);
    D.ghost_perm.reserve((size_t) ng + 1);
    if (ng > 0) {
      D.keys_in.reserve((size_t) ng);
      D.keys_out.reserve((size_t) ng);
      D.vals_in.reserve((size_t) ng);
      D.vals_out.reserve((size_t) ng);
      hipLaunchKernelGGL(k_border_keys, dim3(nblk(ng)), dim3(NB), 0, st, dd, ng, (const HaloRec *) recvbuf,
                         D.keys_in.get(), D.vals_in.get());
      sort_pairs(ctx, D, ng);
      hipLaunchKernelGGL(k_border_finalize, dim3(nblk(ng)), dim3(NB), 0, st, ng, D.vals_out.get(), D.ghost_perm.get());
      hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(ng)), dim3(NB), 0, st, ng, n, D.ghost_perm.get(),
                         (const HaloRec *) recvbuf, ctx->pos4.get(), ctx->meta.get(), ctx->ucgp.get(), ctx->tag.get(), 1);
      hipLaunchKernelGGL(k_bins_from_pos, dim3(nblk(ng)), dim3(NB), 0, st, dd, ng, n, ctx->pos4.get(), D.bin_of.get());
    }
    ctx->nghost = ng;
    ctx->ghost_src_valid = false;  // ghosts of a decomposed run belong to other ranks
    UCG_HIP(hipGetLastError());
    build_bins_and_rows(ctx);
    return UCG_OK;
  });
}

int ucg_halo_pack(ucg_ctx *ctx, void *sendbuf)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    if (D.nsend == 0) return UCG_OK;
    if (!sendbuf) return UCG_ERR_INVALID;
    const DomainDev dd = make_dev(D);
    hipLaunchKernelGGL(k_halo_pack, dim3(nblk(D.nsend)), dim3(NB), 0, ctx->stream, dd, (int) D.nsend, D.send_src.get(),
                       D.send_code.get(), ctx->pos4.get(), ctx->ucgp.get(), ctx->meta.get(), ctx->tag.get(),
                       (HaloRec *) sendbuf, 0);
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int ucg_halo_unpack(ucg_ctx *ctx, const void *recvbuf)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const int ng = ctx->nghost;
    if (ng == 0) return UCG_OK;
    if (!recvbuf) return UCG_ERR_INVALID;
    hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(ng)), dim3(NB), 0, ctx->stream, ng, ctx->nlocal, D.ghost_perm.get(),
                       (const HaloRec *) recvbuf, ctx->pos4.get(), ctx->meta.get(), ctx->ucgp.get(), ctx->tag.get(), 0);
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int ucg_halo_aux_pack(ucg_ctx *ctx, const void *field_dev, void *sendbuf)
{
  if (!ctx || !field_dev) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    if (D.nsend == 0) return UCG_OK;
    if (!sendbuf) return UCG_ERR_INVALID;
    hipLaunchKernelGGL(k_halo_aux_pack, dim3(nblk(D.nsend)), dim3(NB), 0, ctx->stream, (int) D.nsend, D.send_src.get(),
                       (const double2 *) field_dev, (double2 *) sendbuf);
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int ucg_halo_aux_unpack(ucg_ctx *ctx, void *field_dev, const void *recvbuf)
{
  if (!ctx || !field_dev) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    const int ng = ctx->nghost;
    if (ng == 0) return UCG_OK;
    if (!recvbuf) return UCG_ERR_INVALID;
    hipLaunchKernelGGL(k_halo_aux_unpack, dim3(nblk(ng)), dim3(NB), 0, ctx->stream, ng, ctx->nlocal, D.ghost_perm.get(),
                       (const double2 *) recvbuf, (double2 *) field_dev);
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int ucg_halo_molmask_pack(ucg_ctx *ctx, void *sendbuf)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    if (!ctx->has_mol) throw InputError{"no molecule ids were uploaded"};
    if (D.nsend == 0) return UCG_OK;
    if (!sendbuf) return UCG_ERR_INVALID;
    hipLaunchKernelGGL(k_halo_molmask_pack, dim3(nblk(D.nsend)), dim3(NB), 0, ctx->stream, (int) D.nsend,
                       D.send_src.get(), ctx->mask.get(), ctx->mol.get(), (int2 *) sendbuf);
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int ucg_halo_molmask_unpack(ucg_ctx *ctx, const void *recvbuf)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    if (!ctx->has_mol) throw InputError{"no molecule ids were uploaded"};
    const int ng = ctx->nghost;
    if (ng == 0) return UCG_OK;
    if (!recvbuf) return UCG_ERR_INVALID;
    const size_t nall = (size_t) ctx->nlocal + ng;
    ctx->mask.reserve(nall, true, ctx->stream);
    ctx->mol.reserve(nall, true, ctx->stream);
    hipLaunchKernelGGL(k_halo_molmask_unpack, dim3(nblk(ng)), dim3(NB), 0, ctx->stream, ng, ctx->nlocal,
                       D.ghost_perm.get(), (const int2 *) recvbuf, ctx->mask.get(), ctx->mol.get());
    UCG_HIP(hipGetLastError());
    return UCG_OK;
  });
}

int ucg_md_set_timestep(ucg_ctx *ctx, long long ntimestep)
{
  if (!ctx) return UCG_ERR_INVALID;
  ctx->ntimestep = ntimestep;
  return UCG_OK;
}

}  // extern "C"

namespace ucg {
// The re-neighbour decision without its download (RCCL transport: the all-reduce runs on the device values).  *due as
// ucg_decide_local; *checked = 1: the distance check was queued and dev_out3[0] = "a local bead moved > skin / 2",
// dev_out3[1] = "the pair kernels' sticky error flag is up" will be on the stream (dev_out3[2] is the caller's);
// *checked = 0 with *due = 1: no check is made (neigh_modify check no, or fix cluster_switch forces the step): flag = 1.
__global__ void k_flags_any64(int nflags, const int *blockflags, const int *pair_err, long long *out)
{
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  int any = 0;
  for (int b = threadIdx.x; b < nflags; b += blockDim.x) any |= blockflags[b];
  if (__any(any != 0) && (threadIdx.x & 63) == 0) s_any = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = s_any;
    out[1] = (pair_err && *pair_err) ? 1 : 0;
  }
}

int decide_launch(ucg_ctx *ctx, int *due, int *checked, long long *dev_out3)
{
  if (!ctx || !due || !checked || !dev_out3) return UCG_ERR_INVALID;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    Domain &D = *ctx->dom;
    *checked = 0;
    if (cluster_forces_rebuild(ctx)) {
      *due = 1;
      return UCG_OK;
    }
    D.ago++;
    *due = 0;
    if (D.ago >= D.delay && D.ago % D.every == 0) {
      *due = 1;
      if (D.check == 0) return UCG_OK;
      const DomainDev dd = make_dev(D);
      D.blockflags.reserve((size_t) nblk(ctx->nlocal) + 1);
      mirror_need(ctx, UCG_F_X);
      if (ctx->nlocal > 0)
        hipLaunchKernelGGL(k_check_distance, dim3(nblk(ctx->nlocal)), dim3(NB), 0, ctx->stream, dd, ctx->nlocal,
                           ctx->pos4.get(), D.xhold.get(), D.blockflags.get());
      const int *perr = (ctx->md_pair && ctx->md_pair->uploaded) ? ctx->md_pair->d_err.get() : nullptr;
      hipLaunchKernelGGL(k_flags_any64, dim3(1), dim3(1024), 0, ctx->stream, nblk(ctx->nlocal), D.blockflags.get(), perr, dev_out3);
      UCG_HIP(hipGetLastError());
      *checked = 1;
    }
    return UCG_OK;
  });
}

// ucg_decide_local, and -- when the distance check ran -- the pair kernels' sticky error flag in *pair_flag (-1: not read)
int decide_local_impl(ucg_ctx *ctx, int *due, int *flag, int *pair_flag)
{
  if (!ctx || !due || !flag) return UCG_ERR_INVALID;
  if (pair_flag) *pair_flag = -1;
  if (int rc = need_domain(ctx)) return rc;
  return guarded(ctx, [&]() -> int {
    // Neighbor::decide(): the caller combines `flag` over ranks (MPI_Allreduce in upstream)
    Domain &D = *ctx->dom;
    if (cluster_forces_rebuild(ctx)) {  // fix cluster_switch: forced on every rank at the same step
      *due = 1;
      *flag = 1;
      return UCG_OK;
    }
    D.ago++;
    *due = 0;
    *flag = 0;
    if (D.ago >= D.delay && D.ago % D.every == 0) {
      *due = 1;
      if (D.check == 0) {
        *flag = 1;
        return UCG_OK;
      }
      const DomainDev dd = make_dev(D);
      D.blockflags.reserve((size_t) nblk(ctx->nlocal) + 1);
      mirror_need(ctx, UCG_F_X);  // the distance check reads the owned positions
      if (ctx->nlocal > 0)
        hipLaunchKernelGGL(k_check_distance, dim3(nblk(ctx->nlocal)), dim3(NB), 0, ctx->stream, dd, ctx->nlocal,
                           ctx->pos4.get(), D.xhold.get(), D.blockflags.get());
      const int *perr = (pair_flag && ctx->md_pair && ctx->md_pair->uploaded) ? ctx->md_pair->d_err.get() : nullptr;
      D.counter.reserve(4);
      hipLaunchKernelGGL(k_flags_any, dim3(1), dim3(1024), 0, ctx->stream, nblk(ctx->nlocal), D.blockflags.get(),
                         D.counter.get(), perr);
      int f[2] = {0, 0};
      UCG_HIP(hipMemcpyAsync(f, D.counter.get(), 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      UCG_HIP(hipStreamSynchronize(ctx->stream));
      *flag = f[0] != 0;
      if (perr) *pair_flag = f[1];
    }
    return UCG_OK;
  });
}
}  // namespace ucg

extern "C" {

int ucg_decide_local(ucg_ctx *ctx, int *due, int *flag) { return decide_local_impl(ctx, due, flag, nullptr); }

int ucg_record_bytes(int *atom_rec, int *halo_rec)
{
  if (atom_rec) *atom_rec = (int) sizeof(AtomRec);
  if (halo_rec) *halo_rec = (int) sizeof(HaloRec);
  return UCG_OK;
}

}  // extern "C"
