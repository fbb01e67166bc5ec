// How fast does the chip start workgroups?  A kernel of G workgroups x 256 threads whose body idles for `spin` cycles
// (s_sleep loops on the wave's clock), with the LDS / VGPR footprint of a 64x64 igemm tile (4 workgroups per CU) or a
// 128x128 one (2 per CU).  If G x spin / slots is small against the measured time, the time is dispatch.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

struct BigArgs { long long pad[48]; };     // ~384 B of kernel arguments, as IgemmArgs

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void idle_kernel(BigArgs a, int spin, int* sink) {
  __shared__ char lds[LDS_BYTES];
  const long long t0 = clock64();
  if (spin > 0) {
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(2);
  }
  if (a.pad[0] == 12345 && threadIdx.x == 0) { lds[0] = 1; sink[blockIdx.x] = lds[threadIdx.x & 7]; }
}

template <int LDS_BYTES>
float run(int grid, int spin, int* sink, int iters) {
  BigArgs a = {};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(idle_kernel<LDS_BYTES>, dim3(grid), dim3(256), 0, 0, a, spin, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(idle_kernel<LDS_BYTES>, dim3(grid), dim3(256), 0, 0, a, spin, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  int* sink;
  hipMalloc(&sink, 1 << 20);
  const int grids[] = {256, 512, 1024, 2112, 4224, 8448};
  const int spins[] = {0, 2000, 6000, 12000};     // cycles of the 100 MHz..GPU clock counter (clock64 = s_memtime)
  printf("%8s %8s | %10s %10s\n", "wgs", "spin", "32KB us", "64KB us");
  for (int g : grids)
    for (int sp : spins)
      printf("%8d %8d | %10.1f %10.1f\n", g, sp, run<32768>(g, sp, sink, 50), run<65536 - 64>(g, sp, sink, 50));
  return 0;
}
