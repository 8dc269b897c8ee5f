// atomic_rate.hip -- how fast can a half-list scatter be on MI355X?  One lane per bead, 36 neighbours per bead in a
// Morton-local window (+-512 beads), six 64-bit integer atomic adds per neighbour (fx, fy, fz, ucgforce, two scores in
// fixed point): the traffic a deterministic pair-once formulation of the UCG pair loop would generate at 1 M beads.
// build: hipcc -O3 --offload-arch=gfx950 atomic_rate.hip -o atomic_rate ; run: ./atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned hash(unsigned x)
{
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

template <int NFIELD, bool SOA>
__global__ __launch_bounds__(1024) void k_scatter(int n, int nneigh, int window, unsigned long long *acc)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  for (int e = 0; e < nneigh; e++) {
    int m = k + (int) (hash((unsigned) k * 131u + (unsigned) e) % (unsigned) (2 * window + 1)) - window;
    m = m < 0 ? m + n : (m >= n ? m - n : m);
    const unsigned long long v = (unsigned long long) (k + e);
#pragma unroll
    for (int f = 0; f < NFIELD; f++) {
      unsigned long long *p = SOA ? acc + (size_t) f * n + m : acc + (size_t) m * NFIELD + f;
      atomicAdd(p, v + f);
    }
  }
}

template <int NFIELD, bool SOA>
float run(int n, int nneigh, int window, unsigned long long *acc, int reps)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_scatter<NFIELD, SOA>), dim3((n + 1023) / 1024), dim3(1024), 0, 0, n, nneigh, window, acc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_scatter<NFIELD, SOA>), dim3((n + 1023) / 1024), dim3(1024), 0, 0, n, nneigh, window, acc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main()
{
  const int n = 1000000, nneigh = 36;
  unsigned long long *acc;
  CHECK(hipMalloc(&acc, sizeof(unsigned long long) * 6 * (size_t) n));
  CHECK(hipMemset(acc, 0, sizeof(unsigned long long) * 6 * (size_t) n));
  for (int window : {512, 4096}) {
    const float a6 = run<6, false>(n, nneigh, window, acc, 5);
    const float s6 = run<6, true>(n, nneigh, window, acc, 5);
    const float a1 = run<1, false>(n, nneigh, window, acc, 5);
    printf("window +-%d: 6 fields AoS %.1f us (%.1f G atomics/s), SoA %.1f us (%.1f G/s); 1 field %.1f us (%.1f G/s)\n", window,
           a6 * 1e3, 6.0 * n * nneigh / (a6 * 1e-3) / 1e9, s6 * 1e3, 6.0 * n * nneigh / (s6 * 1e-3) / 1e9, a1 * 1e3,
           1.0 * n * nneigh / (a1 * 1e-3) / 1e9);
  }
  return 0;
}
