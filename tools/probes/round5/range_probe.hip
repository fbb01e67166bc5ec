// raw buffer descriptor range check on a dwordx4 load that straddles num_records: per dword or per access?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, unsigned bytes, float* out) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(bytes - 8), 0, 0);     // 2 dwords inside, 2 outside
  const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(bytes - 4), 0, 0);     // 1 inside
  if (threadIdx.x == 0) {
    out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y); out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
    out[4] = __uint_as_float(w.x); out[5] = __uint_as_float(w.y); out[6] = __uint_as_float(w.z); out[7] = __uint_as_float(w.w);
  }
}
int main() {
  const int n = 1024;
  float h[n + 16];
  for (int i = 0; i < n + 16; ++i) h[i] = (float)(i + 1);
  float *d, *o;
  hipMalloc(&d, (n + 16) * 4); hipMalloc(&o, 64);
  hipMemcpy(d, h, (n + 16) * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned)(n * 4), o);
  printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  float r[8];
  hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
  printf("straddle by 2: %g %g %g %g (inside values would be %d %d, outside 0 if checked per dword)\n", r[0], r[1], r[2], r[3], n - 1, n);
  printf("straddle by 3: %g %g %g %g\n", r[4], r[5], r[6], r[7]);
  return 0;
}
