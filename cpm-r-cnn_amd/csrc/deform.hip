// Deformable (v1) and narrow-group 3x3 convolution support for the ResNeXt-64x4d + DCN backbone (BASELINE
// config #5), NHWC, gfx950.
//
// Semantics: pet/lib/ops/csrc/Deformable/deform_conv_cuda_kernel.cu
//   sampling          :95-128  (bilinear on floor(h), floor(w); corners outside the map contribute 0)
//   im2col            :215-287 (h_im = h_out*stride - pad + i*dil + offset_h; sampled only if -1 < h_im < H etc.;
//                               offset channel 2*(i*kw+j) is the row offset, +1 the column offset)
//   col2im (d input)  :290-362 (scatter of the four bilinear corner weights)
//   coord gradient    :365-460 (d offset = sum over the deformable group's channels of col-gradient x d(bilinear)/d(h|w))
//
// Design (not the reference's): the reference materialises columns channel-major and then loops 64 tiny cuBLAS
// GEMMs per image (deform_conv_cuda.cu:402-407).  Here the sampled columns are written pixel-major as
// cols[m][g][tap][c_in_group], which is exactly the NHWC input of a grouped 1x1 convolution whose weight is the
// layer's own KRSC weight -- so the contraction, its data gradient and its weight gradient run on the
// implicit-GEMM MFMA kernels (cpm_conv2d_*), and only the three gather/scatter kernels below are new.  With a NULL
// offset pointer the same kernel is a plain im2col, used for the 4..32-channel groups of ResNeXt's ordinary 3x3s.
// One wave handles one (pixel, tap): the tap's sampling position is wave-uniform, lanes sweep channels, every
// load / store / atomic is channel-contiguous.
#include "common.h"

namespace {

struct DeformGeom {
  int N, H, W, C;      // input [N,H,W,C]
  int P, Q;            // output spatial size
  int R, S, stride, pad, dil;
  int groups, dg;      // conv groups, deformable groups
};

struct Sample {
  float h, w;
  bool valid;
  int h0, w0;
  float lh, lw;
};

__device__ __forceinline__ Sample sample_pos(const DeformGeom& G, const float* __restrict__ offset, int64_t m, int p,
                                             int q, int t, int dgi) {
  Sample s;
  const int i = t / G.S, j = t - i * G.S;
  float oh = 0.f, ow = 0.f;
  if (offset) {
    const float* o = offset + m * (2 * G.R * G.S * G.dg) + (dgi * G.R * G.S + t) * 2;
    oh = o[0];
    ow = o[1];
  }
  s.h = (float)(p * G.stride - G.pad + i * G.dil) + oh;
  s.w = (float)(q * G.stride - G.pad + j * G.dil) + ow;
  s.valid = s.h > -1.f && s.w > -1.f && s.h < (float)G.H && s.w < (float)G.W;
  s.h0 = (int)floorf(s.h);
  s.w0 = (int)floorf(s.w);
  s.lh = s.h - (float)s.h0;
  s.lw = s.w - (float)s.w0;
  return s;
}

// grid: (ceil(M*taps / 4)), block 256 (4 waves, one (pixel, tap) each)
__global__ __launch_bounds__(256) void deform_im2col_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ offset, DeformGeom G,
                                                            float* __restrict__ cols) {
  const int taps = G.R * G.S;
  const int64_t M = (int64_t)G.N * G.P * G.Q;
  const int64_t job = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (job >= M * taps) return;
  const int lane = threadIdx.x & 63;
  const int t = job % taps;
  const int64_t m = job / taps;
  const int q = m % G.Q;
  const int64_t r = m / G.Q;
  const int p = r % G.P, n = r / G.P;
  const int Cg = G.C / G.groups, Cd = G.C / G.dg;
  const float* xb = x + (int64_t)n * G.H * G.W * G.C;
  float* cb = cols + m * (int64_t)taps * G.C;
  for (int dgi = 0; dgi < G.dg; ++dgi) {
    const Sample s = sample_pos(G, offset, m, p, q, t, dgi);
    const float hh = 1.f - s.lh, hw = 1.f - s.lw;
    const float w1 = hh * hw, w2 = hh * s.lw, w3 = s.lh * hw, w4 = s.lh * s.lw;
    const bool v1 = s.valid && s.h0 >= 0 && s.w0 >= 0;
    const bool v2 = s.valid && s.h0 >= 0 && s.w0 + 1 <= G.W - 1;
    const bool v3 = s.valid && s.h0 + 1 <= G.H - 1 && s.w0 >= 0;
    const bool v4 = s.valid && s.h0 + 1 <= G.H - 1 && s.w0 + 1 <= G.W - 1;
    const float* p1 = xb + ((int64_t)s.h0 * G.W + s.w0) * G.C;
    for (int c = dgi * Cd + lane; c < (dgi + 1) * Cd; c += 64) {
      const float a = v1 ? p1[c] : 0.f;
      const float b = v2 ? p1[G.C + c] : 0.f;
      const float d = v3 ? p1[(int64_t)G.W * G.C + c] : 0.f;
      const float e = v4 ? p1[(int64_t)(G.W + 1) * G.C + c] : 0.f;
      const int g = c / Cg, cl = c - g * Cg;
      cb[((int64_t)g * taps + t) * Cg + cl] = w1 * a + w2 * b + w3 * d + w4 * e;
    }
  }
}

// d input: scatter dcols back through the bilinear weights (atomics, channel-contiguous per wave-instruction)
__global__ __launch_bounds__(256) void deform_col2im_kernel(const float* __restrict__ dcols,
                                                            const float* __restrict__ offset, DeformGeom G,
                                                            float* __restrict__ dx) {
  const int taps = G.R * G.S;
  const int64_t M = (int64_t)G.N * G.P * G.Q;
  const int64_t job = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (job >= M * taps) return;
  const int lane = threadIdx.x & 63;
  const int t = job % taps;
  const int64_t m = job / taps;
  const int q = m % G.Q;
  const int64_t r = m / G.Q;
  const int p = r % G.P, n = r / G.P;
  const int Cg = G.C / G.groups, Cd = G.C / G.dg;
  float* xb = dx + (int64_t)n * G.H * G.W * G.C;
  const float* cb = dcols + m * (int64_t)taps * G.C;
  for (int dgi = 0; dgi < G.dg; ++dgi) {
    const Sample s = sample_pos(G, offset, m, p, q, t, dgi);
    if (!s.valid) continue;
    const float hh = 1.f - s.lh, hw = 1.f - s.lw;
    const float w1 = hh * hw, w2 = hh * s.lw, w3 = s.lh * hw, w4 = s.lh * s.lw;
    const bool v1 = s.h0 >= 0 && s.w0 >= 0;
    const bool v2 = s.h0 >= 0 && s.w0 + 1 <= G.W - 1;
    const bool v3 = s.h0 + 1 <= G.H - 1 && s.w0 >= 0;
    const bool v4 = s.h0 + 1 <= G.H - 1 && s.w0 + 1 <= G.W - 1;
    float* p1 = xb + ((int64_t)s.h0 * G.W + s.w0) * G.C;
    for (int c = dgi * Cd + lane; c < (dgi + 1) * Cd; c += 64) {
      const int g = c / Cg, cl = c - g * Cg;
      const float gv = cb[((int64_t)g * taps + t) * Cg + cl];
      // a corner with zero bilinear weight adds nothing: skipping it is exact, and with integer sampling positions
      // (the zero-initialised offset predictor of DeformConvPack, or offset == NULL) it removes 3 of the 4 atomics
      if (v1 && w1 != 0.f) atomicAdd(p1 + c, w1 * gv);
      if (v2 && w2 != 0.f) atomicAdd(p1 + G.C + c, w2 * gv);
      if (v3 && w3 != 0.f) atomicAdd(p1 + (int64_t)G.W * G.C + c, w3 * gv);
      if (v4 && w4 != 0.f) atomicAdd(p1 + (int64_t)(G.W + 1) * G.C + c, w4 * gv);
    }
  }
}

// d offset: per (pixel, tap, deformable group) two numbers, each a reduction over the group's channels
__global__ __launch_bounds__(256) void deform_coord_kernel(const float* __restrict__ dcols,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ offset, DeformGeom G,
                                                           float* __restrict__ doffset) {
  const int taps = G.R * G.S;
  const int64_t M = (int64_t)G.N * G.P * G.Q;
  const int64_t job = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (job >= M * taps) return;
  const int lane = threadIdx.x & 63;
  const int t = job % taps;
  const int64_t m = job / taps;
  const int q = m % G.Q;
  const int64_t r = m / G.Q;
  const int p = r % G.P, n = r / G.P;
  const int Cg = G.C / G.groups, Cd = G.C / G.dg;
  const float* xb = x + (int64_t)n * G.H * G.W * G.C;
  const float* cb = dcols + m * (int64_t)taps * G.C;
  for (int dgi = 0; dgi < G.dg; ++dgi) {
    const Sample s = sample_pos(G, offset, m, p, q, t, dgi);
    float gh = 0.f, gw = 0.f;
    if (s.valid) {
      const bool v1 = s.h0 >= 0 && s.w0 >= 0;
      const bool v2 = s.h0 >= 0 && s.w0 + 1 <= G.W - 1;
      const bool v3 = s.h0 + 1 <= G.H - 1 && s.w0 >= 0;
      const bool v4 = s.h0 + 1 <= G.H - 1 && s.w0 + 1 <= G.W - 1;
      const float* p1 = xb + ((int64_t)s.h0 * G.W + s.w0) * G.C;
      const float hw = 1.f - s.lw, hh = 1.f - s.lh;
      for (int c = dgi * Cd + lane; c < (dgi + 1) * Cd; c += 64) {
        const float a = v1 ? p1[c] : 0.f;
        const float b = v2 ? p1[G.C + c] : 0.f;
        const float d = v3 ? p1[(int64_t)G.W * G.C + c] : 0.f;
        const float e = v4 ? p1[(int64_t)(G.W + 1) * G.C + c] : 0.f;
        const int g = c / Cg, cl = c - g * Cg;
        const float gv = cb[((int64_t)g * taps + t) * Cg + cl];
        // deform_conv_cuda_kernel.cu:185-209: d/dh = -(hw a + lw b) + (hw d + lw e), d/dw = -(hh a) + hh b - lh d + lh e
        gh += gv * (-(hw * a) - s.lw * b + hw * d + s.lw * e);
        gw += gv * (-(hh * a) + hh * b - s.lh * d + s.lh * e);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      gh += __shfl_xor(gh, o, 64);
      gw += __shfl_xor(gw, o, 64);
    }
    if (lane == 0) {
      float* o = doffset + m * (2 * taps * G.dg) + (dgi * taps + t) * 2;
      o[0] = gh;
      o[1] = gw;
    }
  }
}

int check_geom(const DeformGeom& G) {
  if (G.N <= 0 || G.H <= 0 || G.W <= 0 || G.C <= 0 || G.R <= 0 || G.S <= 0 || G.stride <= 0 || G.dil <= 0) return -1;
  if (G.groups <= 0 || G.dg <= 0 || G.C % G.groups || G.C % G.dg) return -1;
  if (G.P != (G.H + 2 * G.pad - G.dil * (G.R - 1) - 1) / G.stride + 1) return -1;
  if (G.Q != (G.W + 2 * G.pad - G.dil * (G.S - 1) - 1) / G.stride + 1) return -1;
  return 0;
}

DeformGeom make_geom(int N, int H, int W, int C, int R, int S, int stride, int pad, int dil, int groups, int dg, int P,
                     int Q) {
  DeformGeom G;
  G.N = N; G.H = H; G.W = W; G.C = C; G.P = P; G.Q = Q; G.R = R; G.S = S;
  G.stride = stride; G.pad = pad; G.dil = dil; G.groups = groups; G.dg = dg;
  return G;
}

unsigned jobs_grid(const DeformGeom& G) {
  const int64_t jobs = (int64_t)G.N * G.P * G.Q * G.R * G.S;
  return (unsigned)((jobs + 3) / 4);
}

}  // namespace

CPM_EXPORT int cpm_deform_im2col(const float* x, const float* offset, int N, int H, int W, int C, int R, int S,
                                 int stride, int pad, int dilation, int groups, int deformable_groups, int P, int Q,
                                 float* cols, void* stream) {
  DeformGeom G = make_geom(N, H, W, C, R, S, stride, pad, dilation, groups, deformable_groups, P, Q);
  CPM_REQUIRE(check_geom(G) == 0, "bad geometry");
  CPM_REQUIRE(x && cols, "null pointer");
  hipLaunchKernelGGL(deform_im2col_kernel, dim3(jobs_grid(G)), dim3(256), 0, (hipStream_t)stream, x, offset, G, cols);
  return cpm::check_launch("deform_im2col");
}

CPM_EXPORT int cpm_deform_col2im(const float* dcols, const float* offset, int N, int H, int W, int C, int R, int S,
                                 int stride, int pad, int dilation, int groups, int deformable_groups, int P, int Q,
                                 float* dx, void* stream) {
  DeformGeom G = make_geom(N, H, W, C, R, S, stride, pad, dilation, groups, deformable_groups, P, Q);
  CPM_REQUIRE(check_geom(G) == 0, "bad geometry");
  CPM_REQUIRE(dcols && dx, "null pointer");
  hipLaunchKernelGGL(deform_col2im_kernel, dim3(jobs_grid(G)), dim3(256), 0, (hipStream_t)stream, dcols, offset, G,
                     dx);
  return cpm::check_launch("deform_col2im");
}

CPM_EXPORT int cpm_deform_coord_grad(const float* dcols, const float* x, const float* offset, int N, int H, int W,
                                     int C, int R, int S, int stride, int pad, int dilation, int groups,
                                     int deformable_groups, int P, int Q, float* doffset, void* stream) {
  DeformGeom G = make_geom(N, H, W, C, R, S, stride, pad, dilation, groups, deformable_groups, P, Q);
  CPM_REQUIRE(check_geom(G) == 0, "bad geometry");
  CPM_REQUIRE(dcols && x && offset && doffset, "null pointer");
  hipLaunchKernelGGL(deform_coord_kernel, dim3(jobs_grid(G)), dim3(256), 0, (hipStream_t)stream, dcols, x, offset, G,
                     doffset);
  return cpm::check_launch("deform_coord_grad");
}
