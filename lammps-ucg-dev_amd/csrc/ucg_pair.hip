// ucg_pair.hip -- launchers of the neighbour-loop kernels of table_ucgld and table_ucg_bethe (gfx950) and the
// instantiations of k_pair_gather (ucg_pair_kernel.h) other than the tuned ones of ucg_pair_hot.hip: tables through L1 / L2
// (with or without an LDS hot block), per-table r^2 grids, BITMAP tables, 4 / 8 / 16 lanes per bead.
#include "ucg_pair_kernel.h"

namespace ucg {

// ucg_pair_hot.hip
hipError_t launch_pair_gather_hot(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart, int *errflag,
                                  hipStream_t st, int nblocks);
hipError_t launch_pair_gather_hot_fused(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart, int *errflag,
                                        hipStream_t st, int nblocks);

namespace {

// the variants ucg_pair_hot.hip holds: tables in LDS on one shared r^2 grid, one or two lanes per bead
inline bool pair_gather_is_hot(const PairDev &P)
{
  return P.tab_in_lds && P.fast && P.tabstyle != 3 && (P.gather_slots == 1 || P.gather_slots == 2);
}

// per workgroup of the gather kernel: does any of its beads have a ghost among its neighbours?
__global__ __launch_bounds__(256) void k_block_classify(int nlocal, int beads_per_block, int pitch, const int *numneigh,
                                                       const int *neigh, int *flags)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nlocal) return;
  const int n = numneigh[k];
  bool ghost = false;
  for (int e = 0; e < n; e++) ghost |= (neigh[(size_t) e * pitch + k] & 0x1FFFFFFF) >= nlocal;
  if (ghost) flags[k / beads_per_block] = 1;
}

__global__ void k_ev_final(const double *part, int nblocks, double *out)
{
  // fixed-order sum of the per-block partials
  const int c = threadIdx.x;
  if (c < 8) {
    double s = 0.0;
    for (int b = 0; b < nblocks; b++) s += part[(size_t) b * 8 + c];
    out[c] = s;
  }
}

// exactness check of div_by_const against the IEEE division on random operands
__global__ void k_selftest_div(const double b, const double y, const unsigned long long seed, const int n,
                               unsigned long long *mismatches)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // splitmix64 -> a double with a random sign, exponent in [-40, 40] and random significand
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long) (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const unsigned long long mant = z & 0xFFFFFFFFFFFFFull;
  const unsigned long long ex = 1023ull - 40ull + ((z >> 52) % 81ull);
  const unsigned long long sg = (z >> 63) << 63;
  const double a = __builtin_bit_cast(double, sg | (ex << 52) | mant);
  const double q1 = a / b;
  const double q2 = div_by_const(a, b, y);
  if (__builtin_bit_cast(unsigned long long, q1) != __builtin_bit_cast(unsigned long long, q2)) atomicAdd(mismatches, 1ull);
}

// exactness check of ucg_div_core (ucg_math.h) against the IEEE division on the operand ranges it is used on: denominators
// in [1.25, 2.75] and [5.25, 6.75], numerators with a random sign, significand and exponent in [-112, 2], and zero
__global__ void k_selftest_div_core(const unsigned long long seed, const int n, unsigned long long *mismatches)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long) (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  unsigned long long w = z * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  w = (w ^ (w >> 29)) * 0xBF58476D1CE4E5B9ull;
  w ^= w >> 32;
  const unsigned long long mant = z & 0xFFFFFFFFFFFFFull;
  const unsigned long long ex = 1023ull - 112ull + ((z >> 52) % 115ull);
  const unsigned long long sg = (z >> 63) << 63;
  double a = __builtin_bit_cast(double, sg | (ex << 52) | mant);
  if ((w & 0xFFFull) == 0) a = 0.0;
  const double u = (double) (w >> 11) * 1.1102230246251565e-16;  // [0, 1)
  const double b = ((w >> 3) & 1ull) ? 1.25 + 1.5 * u : 5.25 + 1.5 * u;
  const double q1 = a / b;
  const double q2 = ucg_div_core(a, b);
  if (__builtin_bit_cast(unsigned long long, q1) != __builtin_bit_cast(unsigned long long, q2)) atomicAdd(mismatches, 1ull);
}

// exactness check of ucg_sqrt_core (ucg_math.h) against sqrt on the range it is used on: random significands, exponents
// -700 ... 699 (and the perfect squares and their neighbours in the last place, where the rounding is decided by one bit)
__global__ void k_selftest_sqrt_core(const unsigned long long seed, const int n, unsigned long long *mismatches)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long) (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const unsigned long long mant = z & 0xFFFFFFFFFFFFFull;
  const unsigned long long ex = 1023ull - 700ull + ((z >> 52) % 1400ull);
  double x = __builtin_bit_cast(double, (ex << 52) | mant);
  if ((i & 7) == 7) {
    // a square of a 26-bit significand (exactly representable), or its neighbour one unit in the last place up / down
    const double m = (double) ((z >> 8) & 0x3FFFFFFull) + 1.0;
    const double sq = m * m;
    const unsigned long long b = __builtin_bit_cast(unsigned long long, sq) + (unsigned long long) ((long long) ((z >> 40) % 3ull) - 1ll);
    x = __builtin_bit_cast(double, b);
  }
  const double q1 = sqrt(x);
  const double q2 = ucg_sqrt_core(x);
  if (__builtin_bit_cast(unsigned long long, q1) != __builtin_bit_cast(unsigned long long, q2)) atomicAdd(mismatches, 1ull);
}

template <int STYLE, int TS, int SLOTS>
hipError_t launch_style_ts(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart,
                           int *errflag, hipStream_t st, int nblocks)
{
  const size_t tabbytes = P.fast ? ((size_t) (P.tablength * P.fast_stride + 1) / 2) * sizeof(double4)
                                 : (size_t) P.ntab * P.tablength * sizeof(double4);
  const size_t ownbytes = P.stage_own ? (size_t) (PAIR_BLOCK / SLOTS) * pair_own_bytes(STYLE) : 0;
  const size_t ldsbytes = (P.tab_in_lds ? tabbytes : (P.fast ? (size_t) P.hot_ent * sizeof(double4) : 0)) + ownbytes;
#define UCG_LAUNCH(EVF, LDSF, FASTF)                                                                   \
  do {                                                                                                 \
    if constexpr (LDSF && FASTF && SLOTS <= 2) {                                                       \
      return hipErrorInvalidValue; /* ucg_pair_hot.hip's (launch_pair_gather dispatches them) */       \
    } else {                                                                                           \
      auto kern = k_pair_gather<STYLE, TS, EVF, LDSF, FASTF, SLOTS>;                                   \
      if (ldsbytes > 48 * 1024) {                                                                      \
        hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int) ldsbytes);                                            \
        if (e != hipSuccess) return e;                                                                 \
      }                                                                                                \
      hipLaunchKernelGGL(kern, dim3(nblocks), dim3(PAIR_BLOCK), ldsbytes, st, P, A, L, evpart, errflag); \
    }                                                                                                  \
  } while (0)
  const int sel = (P.tab_in_lds ? 4 : 0) | (ev ? 2 : 0) | (P.fast ? 1 : 0);
  switch (sel) {
    case 0: UCG_LAUNCH(false, false, false); break;
    case 1: UCG_LAUNCH(false, false, true); break;
    case 2: UCG_LAUNCH(true, false, false); break;
    case 3: UCG_LAUNCH(true, false, true); break;
    case 4: UCG_LAUNCH(false, true, false); break;
    case 5: UCG_LAUNCH(false, true, true); break;
    case 6: UCG_LAUNCH(true, true, false); break;
    default: UCG_LAUNCH(true, true, true); break;
  }
#undef UCG_LAUNCH
  return hipGetLastError();
}

// BITMAP tables: generic path only (bins in HBM / L2, per-table parameters), one lane per bead
template <int STYLE>
hipError_t launch_style_bitmap(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart,
                               int *errflag, hipStream_t st, int nblocks)
{
  if (P.tab_in_lds || P.fast || P.gather_slots != 1) return hipErrorInvalidValue;
  const size_t ldsbytes = P.stage_own ? (size_t) PAIR_BLOCK * pair_own_bytes(STYLE) : 0;
  if (ev) hipLaunchKernelGGL((k_pair_gather<STYLE, 3, true, false, false, 1>), dim3(nblocks), dim3(PAIR_BLOCK), ldsbytes, st, P, A, L, evpart, errflag);
  else hipLaunchKernelGGL((k_pair_gather<STYLE, 3, false, false, false, 1>), dim3(nblocks), dim3(PAIR_BLOCK), ldsbytes, st, P, A, L, evpart, errflag);
  return hipGetLastError();
}

template <int STYLE>
hipError_t launch_style(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart,
                        int *errflag, hipStream_t st, int nblocks)
{
  if (P.tabstyle == 3) return launch_style_bitmap<STYLE>(P, A, L, ev, evpart, errflag, st, nblocks);
#define UCG_TS(SL)                                                                                  \
  switch (P.tabstyle) {                                                                             \
    case 0: return launch_style_ts<STYLE, 0, SL>(P, A, L, ev, evpart, errflag, st, nblocks);       \
    case 1: return launch_style_ts<STYLE, 1, SL>(P, A, L, ev, evpart, errflag, st, nblocks);       \
    default: return launch_style_ts<STYLE, 2, SL>(P, A, L, ev, evpart, errflag, st, nblocks);      \
  }
  switch (P.gather_slots) {
    case 2: UCG_TS(2)
    case 4: UCG_TS(4)
    case 8: UCG_TS(8)
    case 16: UCG_TS(16)
    default: UCG_TS(1)
  }
#undef UCG_TS
}

}  // namespace

#ifdef UCG_FUSED
#define launch_pair_gather launch_pair_gather_fused
#endif

#ifndef UCG_FUSED
int pair_gather_blocks(int nlocal, int slots)
{
  return (int) (((long long) nlocal * slots + PAIR_BLOCK - 1) / PAIR_BLOCK);
}
#endif

hipError_t launch_pair_gather(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev,
                              double *evpart, double *evout, int *errflag, hipStream_t st)
{
  const int nblocks = pair_gather_blocks(A.nlocal, P.gather_slots);
  if (nblocks == 0) return hipSuccess;
  hipError_t e;
#ifdef UCG_FUSED
  if (pair_gather_is_hot(P)) e = launch_pair_gather_hot_fused(P, A, L, ev, evpart, errflag, st, nblocks);
#else
  if (pair_gather_is_hot(P)) e = launch_pair_gather_hot(P, A, L, ev, evpart, errflag, st, nblocks);
#endif
  else if (P.style == 0) e = launch_style<0>(P, A, L, ev, evpart, errflag, st, nblocks);
  else e = launch_style<1>(P, A, L, ev, evpart, errflag, st, nblocks);
  if (e != hipSuccess) return e;
  if (ev) {
    hipLaunchKernelGGL(k_ev_final, dim3(1), dim3(64), 0, st, evpart, nblocks, evout);
    e = hipGetLastError();
  }
  return e;
}

#ifndef UCG_FUSED
hipError_t launch_ev_final(const double *part, int nblocks, double *out, hipStream_t st)
{
  hipLaunchKernelGGL(k_ev_final, dim3(1), dim3(64), 0, st, part, nblocks, out);
  return hipGetLastError();
}

hipError_t launch_block_classify(const AtomsDev &A, const ListDev &L, int slots, int *flags, hipStream_t st)
{
  const int nblocks = pair_gather_blocks(A.nlocal, slots);
  if (nblocks == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(flags, 0, (size_t) nblocks * sizeof(int), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_block_classify, dim3((A.nlocal + 255) / 256), dim3(256), 0, st, A.nlocal, PAIR_BLOCK / slots, L.pitch,
                     L.numneigh, L.neigh, flags);
  return hipGetLastError();
}

hipError_t launch_selftest_sqrt_core(unsigned long long seed, int n, unsigned long long *d_mismatches, hipStream_t st)
{
  hipLaunchKernelGGL(k_selftest_sqrt_core, dim3((n + 255) / 256), dim3(256), 0, st, seed, n, d_mismatches);
  return hipGetLastError();
}

hipError_t launch_selftest_div_core(unsigned long long seed, int n, unsigned long long *d_mismatches, hipStream_t st)
{
  hipLaunchKernelGGL(k_selftest_div_core, dim3((n + 255) / 256), dim3(256), 0, st, seed, n, d_mismatches);
  return hipGetLastError();
}

hipError_t launch_selftest_div(double b, unsigned long long seed, int n, unsigned long long *d_mismatches, hipStream_t st)
{
  hipLaunchKernelGGL(k_selftest_div, dim3((n + 255) / 256), dim3(256), 0, st, b, 1.0 / b, seed, n, d_mismatches);
  return hipGetLastError();
}

#endif  // !UCG_FUSED

}  // namespace ucg
