// ucg_pair_hot.hip -- the tuned instantiations of k_pair_gather (ucg_pair_kernel.h): tables in LDS on one shared r^2 grid
// (LDS_TAB, FAST), one or two lanes per bead.  They are what every headline configuration runs; kept in a translation unit
// of their own so that work on the hot kernel rebuilds in seconds (the remaining ~200 variants live in ucg_pair.hip).
// Compiled a second time with -ffp-contract=fast -DUCG_FUSED for option "fma_contract" (never the bit-exact path).
#include "ucg_pair_kernel.h"

namespace ucg {

namespace {

template <int STYLE, int TS, int SLOTS>
hipError_t launch_hot_ts(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart, int *errflag,
                         hipStream_t st, int nblocks)
{
  const size_t tabbytes = ((size_t) (P.tablength * P.fast_stride + 1) / 2) * sizeof(double4);
  const size_t ownbytes = P.stage_own ? (size_t) (PAIR_BLOCK / SLOTS) * pair_own_bytes(STYLE) : 0;
  const size_t ldsbytes = tabbytes + ownbytes;
#define UCG_LAUNCH(EVF)                                                                                 \
  do {                                                                                                  \
    auto kern = k_pair_gather<STYLE, TS, EVF, true, true, SLOTS>;                                       \
    const bool p2 = P.kT_pow2 != 0;                                                                     \
    if constexpr (STYLE == 1) {                                                                         \
      if (P.onetype_same10 && !P.first_possible) /* (the variants without the first-call rules) */      \
        kern = P.pseudo_flag ? (p2 ? k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 1, 1>       \
                                   : k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 1>)         \
                             : (p2 ? k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 0, 1>       \
                                   : k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 0>);        \
    } else if (P.onetype_same10) {                                                                      \
      kern = p2 ? k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, -1, 1>                         \
                : k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true>;                               \
    }                                                                                                   \
    if constexpr (!EVF) {                                                                               \
      /* lists larger than the last-level cache: the variants that stream their rows (the tuned one-type kernels) */ \
      if (L.stream_rows && P.onetype_same10) {                                                          \
        if constexpr (STYLE == 1) {                                                                     \
          if (!P.first_possible)                                                                        \
            kern = P.pseudo_flag ? (p2 ? k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 1, 1, true>   \
                                       : k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 1, -1, true>) \
                                 : (p2 ? k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 0, 1, true>   \
                                       : k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, 0, -1, true>); \
        } else {                                                                                        \
          kern = p2 ? k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, -1, 1, true>               \
                    : k_pair_gather<STYLE, TS, EVF, true, true, SLOTS, true, -1, -1, true>;             \
        }                                                                                               \
      }                                                                                                 \
    }                                                                                                   \
    if (ldsbytes > 48 * 1024) {                                                                         \
      hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int) ldsbytes);                                               \
      if (e != hipSuccess) return e;                                                                    \
    }                                                                                                   \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(PAIR_BLOCK), ldsbytes, st, P, A, L, evpart, errflag);  \
  } while (0)
  if (ev) UCG_LAUNCH(true);
  else UCG_LAUNCH(false);
#undef UCG_LAUNCH
  return hipGetLastError();
}

template <int STYLE>
hipError_t launch_hot_style(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart, int *errflag,
                            hipStream_t st, int nblocks)
{
#define UCG_TS(SL)                                                                              \
  switch (P.tabstyle) {                                                                         \
    case 0: return launch_hot_ts<STYLE, 0, SL>(P, A, L, ev, evpart, errflag, st, nblocks);      \
    case 1: return launch_hot_ts<STYLE, 1, SL>(P, A, L, ev, evpart, errflag, st, nblocks);      \
    default: return launch_hot_ts<STYLE, 2, SL>(P, A, L, ev, evpart, errflag, st, nblocks);     \
  }
  if (P.gather_slots == 2) { UCG_TS(2) }
  UCG_TS(1)
#undef UCG_TS
}

}  // namespace

#ifdef UCG_FUSED
#define launch_pair_gather_hot launch_pair_gather_hot_fused
#endif

// the caller (ucg_pair.hip: launch_pair_gather) has checked pair_gather_is_hot(P)
hipError_t launch_pair_gather_hot(const PairDev &P, const AtomsDev &A, const ListDev &L, bool ev, double *evpart, int *errflag,
                                  hipStream_t st, int nblocks)
{
  if (P.style == 0) return launch_hot_style<0>(P, A, L, ev, evpart, errflag, st, nblocks);
  return launch_hot_style<1>(P, A, L, ev, evpart, errflag, st, nblocks);
}

}  // namespace ucg
