// Does gfx950 (as this image's driver configures it) serve 16-byte vector loads at 4-/8-byte-aligned addresses?
// buffer_load_dwordx4 through a raw descriptor and global_load_dwordx4, offsets 0, 4, 8, 12 bytes off a 16-byte boundary.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, unsigned bytes, float* out) {
  const int lane = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  for (int sh = 0; sh < 4; ++sh) {
    const unsigned off = (unsigned)(lane * 72 + sh * 4);          // rows of 18 floats, like a [M][18] tensor
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    float4 g;
    asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(g) : "v"((const char*)src + off) : "memory");
    float* o = out + (sh * 64 + lane) * 8;
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    o[4] = g.x; o[5] = g.y; o[6] = g.z; o[7] = g.w;
  }
}
int main() {
  const int n = 64 * 18 + 64;
  float* h = (float*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 4 * 64 * 8 * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned)(n * 4), o);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  float* r = (float*)malloc(4 * 64 * 8 * 4);
  hipMemcpy(r, o, 4 * 64 * 8 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int sh = 0; sh < 4; ++sh)
    for (int l = 0; l < 64; ++l)
      for (int k = 0; k < 4; ++k) {
        const float want = (float)(l * 18 + sh + k);
        if (r[(sh * 64 + l) * 8 + k] != want || r[(sh * 64 + l) * 8 + 4 + k] != want) ++bad;
      }
  printf("mismatches: %d of %d\n", bad, 4 * 64 * 4);
  return bad != 0;
}
