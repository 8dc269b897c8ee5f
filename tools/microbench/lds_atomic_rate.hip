// lds_atomic_rate.hip -- what do order-free LDS accumulators cost on MI355X?  One 1024-lane workgroup per CU; every lane
// adds six 64-bit values (fx, fy, fz, ucgforce, two scores in fixed point) per iteration to the LDS accumulators of a
// pseudo-random bead of its workgroup (NB beads): the traffic the "own-block pairs once" form of the UCG pair loop
// would put on the LDS next to its knot reads.  Reports LDS cycles per wave-level atomic instruction per CU.
// build: hipcc -O3 --offload-arch=gfx950 lds_atomic_rate.hip -o lds_atomic_rate ; run: ./lds_atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned hash(unsigned x)
{
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// MODE 0: 64-bit integer add, SoA (field-major); 1: 64-bit integer add, AoS (bead-major, 48-byte records);
// 2: double add (ds_add_f64), SoA; 3: plain ds_read_b128 of a random 16-byte slot (the knot reads, for scale);
// 4: AoS with 56-byte (odd 8-byte stride) records
template <int MODE, int NFIELD>
__global__ __launch_bounds__(1024) void k_lds(int nb, int iters, unsigned long long *out)
{
  extern __shared__ unsigned long long acc[];
  const int stride = MODE == 4 ? 7 : NFIELD;
  for (int t = threadIdx.x; t < nb * stride; t += blockDim.x) acc[t] = 0;
  __syncthreads();
  unsigned h = hash(blockIdx.x * 1024u + threadIdx.x);
  unsigned long long sink = 0;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    const int m = (int) ((h >> 8) % (unsigned) nb);
    if (MODE == 3) {
      const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(acc);
#pragma unroll
      for (int f = 0; f < NFIELD; f++) {
        const ulonglong2 v = p[(m * 7 + f) % (nb * NFIELD / 2)];
        sink += v.x ^ v.y;
      }
    } else {
#pragma unroll
      for (int f = 0; f < NFIELD; f++) {
        if (MODE == 0) atomicAdd(&acc[f * nb + m], (unsigned long long) (it + f));
        else if (MODE == 1 || MODE == 4) atomicAdd(&acc[m * stride + f], (unsigned long long) (it + f));
        else atomicAdd(reinterpret_cast<double *>(&acc[f * nb + m]), (double) (it + f));
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nb) sink += acc[threadIdx.x];
  if (sink == 0x1234567ull) out[0] = sink;
}

template <int MODE, int NFIELD>
void run(const char *name, int nb, int iters, unsigned long long *out, int nblocks)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t lds = (size_t) nb * 7 * 8;
  CHECK(hipFuncSetAttribute((const void *) k_lds<MODE, NFIELD>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipLaunchKernelGGL((k_lds<MODE, NFIELD>), dim3(nblocks), dim3(1024), lds, 0, nb, iters, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_lds<MODE, NFIELD>), dim3(nblocks), dim3(1024), lds, 0, nb, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double waveinstr_per_cu = 16.0 * iters * NFIELD * (nblocks / 256.0);
  printf("%-44s nb=%4d: %8.1f us, %6.1f ns per wave-instr per CU = %5.1f cycles at 2.4 GHz\n", name, nb, ms * 1e3,
         ms * 1e6 / waveinstr_per_cu, ms * 1e6 / waveinstr_per_cu * 2.4);
}

int main()
{
  unsigned long long *out;
  CHECK(hipMalloc(&out, 64));
  const int iters = 2000, nblocks = 256;
  for (int nb : {512, 1024}) {
    run<0, 6>("ds_add_u64, 6 fields, SoA", nb, iters, out, nblocks);
    run<1, 6>("ds_add_u64, 6 fields, AoS 48 B", nb, iters, out, nblocks);
    run<4, 6>("ds_add_u64, 6 fields, AoS 56 B", nb, iters, out, nblocks);
    run<2, 6>("ds_add_f64, 6 fields, SoA", nb, iters, out, nblocks);
    run<3, 6>("ds_read_b128, 6 random-record slots", nb, iters, out, nblocks);
  }
  return 0;
}
