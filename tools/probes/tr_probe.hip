// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/tr_probe tools/probes/tr_probe.hip ; run on the GPU box: prints, per lane,
// which (row, column) of a [row][col] 16-bit LDS image each element of ds_read_b64_tr_b16 delivers (used for wgrad_taps_kernel)
// probe of ds_read_b64_tr_b16 (gfx950): which (row, column) of a [row][col] 16-bit LDS image lands in which lane element
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16* lds_v4i16_p;
__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[64][64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i / 64][i % 64] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
  const unsigned short* src = &img[4 * g + q][4 * p];
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_p)src);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; ++lane) {
    printf("lane %2d:", lane);
    for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[lane * 4 + e] >> 8, h[lane * 4 + e] & 255);
    printf("\n");
  }
  return 0;
}
