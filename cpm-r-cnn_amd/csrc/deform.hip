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
#include <stdlib.h>

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

__device__ __forceinline__ Sample sample_at(const DeformGeom& G, int p, int q, int t, float oh, float ow) {
  Sample s;
  const int i = t / G.S, j = t - i * G.S;
  s.h = (float)(p * G.stride - G.pad + i * G.dil) + oh;
  s.w = (float)(q * G.stride - G.pad + j * G.dil) + ow;
  s.valid = s.h > -1.f && s.w > -1.f && s.h < (float)G.H && s.w < (float)G.W;
  s.h0 = (int)floorf(s.h);
  s.w0 = (int)floorf(s.w);
  s.lh = s.h - (float)s.h0;
  s.lw = s.w - (float)s.w0;
  return s;
}

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


// d input, tiled.  A workgroup owns a TxT patch of OUTPUT pixels and a slab of 64 channels.  The bilinear corners of
// the patch's R*S*T*T samples land in a small window of input pixels around the patch; the workgroup accumulates them
// there in LDS and adds the window to dx once per cell: a 3x3 stride-1 layer sends 36 corner contributions to every
// input element, routed through the window they reach memory as ~6 float atomics per output pixel -- the scatter
// kernel above runs AT the chip's float-atomic rate (~1.3 TB/s of added bytes, MI355X_MICROARCH.md), this one moves
// a sixth of its traffic.  No LDS atomics: wave w owns channels [16w, 16w+16) of the slab, and its 64 lanes are
// 16 channels x the sample's FOUR corners, so one ds_read / fma / ds_write updates all four corners of a sample;
// cells of different corners are distinct, channels of different waves are distinct, and the LDS operations of one
// wave execute in order.  Samples whose offsets carry them outside the window (margin MG input pixels around the
// patch's nominal footprint) take the direct atomic.
constexpr int COL2IM_T = 4, COL2IM_MG = 2, COL2IM_CP = 80;     // cell pitch in floats: corner groups on distinct banks

__global__ __launch_bounds__(256) void deform_col2im_tiled_kernel(const float* __restrict__ dcols,
                                                                  const float* __restrict__ offset, DeformGeom G,
                                                                  float* __restrict__ dx, int win_h, int win_w) {
  extern __shared__ float win[];                       // [win_h][win_w][COL2IM_CP]
  // one record per sample of the patch, computed ONCE (thread per sample) instead of by every wave in the loop:
  // x = window cell of corner 0 (-1: outside the window -> direct atomics, -2: sample off the map), y / z = lh / lw,
  // w = (h0 + 2^15) << 16 | (w0 + 2^15) | corner-on-map bits in ... (kept separately in smask)
  __shared__ int4 srec[COL2IM_T * COL2IM_T][16];
  __shared__ int smask[COL2IM_T * COL2IM_T][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int taps = G.R * G.S;
  const int tiles_q = (G.Q + COL2IM_T - 1) / COL2IM_T, tiles_p = (G.P + COL2IM_T - 1) / COL2IM_T;
  int b = blockIdx.x;
  const int tq = b % tiles_q; b /= tiles_q;
  const int tp = b % tiles_p;
  const int n = b / tiles_p;
  const int c0 = blockIdx.y * 64;
  const int corner = lane >> 4;                        // 0: (h0, w0)  1: (h0, w0+1)  2: (h0+1, w0)  3: (h0+1, w0+1)
  const int cw = wave * 16 + (lane & 15);              // channel inside the slab
  const int c = c0 + cw;
  const bool c_ok = c < G.C;
  const int Cg = G.C / G.groups, Cd = G.C / G.dg;
  const int cc = c_ok ? c : 0;
  const int g = cc / Cg, cl = cc - g * Cg;
  const int dgu = __builtin_amdgcn_readfirstlane(c0 / Cd);     // uniform: the host takes this path when Cd % 64 == 0
  const int p0 = tp * COL2IM_T, q0 = tq * COL2IM_T;
  const int wy0 = p0 * G.stride - G.pad - COL2IM_MG, wx0 = q0 * G.stride - G.pad - COL2IM_MG;   // window origin
  const int cells = win_h * win_w;
  for (int idx = tid; idx < COL2IM_T * COL2IM_T * taps; idx += 256) {
    const int px = idx / taps, t = idx - px * taps;
    const int p = p0 + px / COL2IM_T, q = q0 + px % COL2IM_T;
    int4 rec = make_int4(-2, 0, 0, 0);
    int mask = 0;
    if (p < G.P && q < G.Q) {
      float2 o = make_float2(0.f, 0.f);
      if (offset) {
        const int64_t m = ((int64_t)n * G.P + p) * G.Q + q;
        o = *(const float2*)(offset + m * (2 * taps * G.dg) + (dgu * taps + t) * 2);
      }
      const Sample sm_ = sample_at(G, p, q, t, o.x, o.y);
      if (sm_.valid) {
        const int wy = sm_.h0 - wy0, wx = sm_.w0 - wx0;
        const bool inside = wy >= 0 && wx >= 0 && wy + 1 < win_h && wx + 1 < win_w;
        rec.x = inside ? wy * win_w + wx : -1;
        rec.y = __float_as_int(sm_.lh);
        rec.z = __float_as_int(sm_.lw);
        rec.w = (int)(((unsigned)(sm_.h0 + 32768) << 16) | ((unsigned)(sm_.w0 + 32768) & 0xFFFFu));
        for (int k = 0; k < 4; ++k) {
          const int y = sm_.h0 + (k >> 1), x = sm_.w0 + (k & 1);
          if ((unsigned)y < (unsigned)G.H && (unsigned)x < (unsigned)G.W) mask |= 1 << k;
        }
      }
    }
    srec[px][t] = rec;
    smask[px][t] = mask;
  }
  for (int i = tid; i < cells * COL2IM_CP; i += 256) win[i] = 0.f;
  __syncthreads();
  float* xb = dx + (int64_t)n * G.H * G.W * G.C;
  const int dy = corner >> 1, dxw = corner & 1;
  const int my_cell_off = (dy * win_w + dxw) * COL2IM_CP + cw;
  auto load_px = [&](int px, float (&gv)[16]) {
    const int p = p0 + px / COL2IM_T, q = q0 + px % COL2IM_T;
    const bool ok = c_ok && px < COL2IM_T * COL2IM_T && p < G.P && q < G.Q;
    const int64_t m = ((int64_t)n * G.P + (ok ? p : 0)) * G.Q + (ok ? q : 0);
    const float* cb = dcols + m * (int64_t)taps * G.C + (int64_t)g * taps * Cg + cl;
#pragma unroll
    for (int t = 0; t < 16; ++t) gv[t] = (t < taps && ok) ? cb[t * Cg] : 0.f;
  };
  auto scatter_px = [&](int px, const float (&gv)[16]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t >= taps) break;
      const int4 rec = srec[px][t];
      if (rec.x == -2) continue;                                          // uniform
      const float lh = __int_as_float(rec.y), lw = __int_as_float(rec.z);
      const float wgt = (dy ? lh : 1.f - lh) * (dxw ? lw : 1.f - lw);
      const bool on_map = (smask[px][t] >> corner) & 1;
      const float add = on_map ? wgt * gv[t] : 0.f;
      if (rec.x >= 0) {                                                   // uniform: inside the window
        win[rec.x * COL2IM_CP + my_cell_off] += add;                      // (a zero add keeps the update branch-free)
      } else if (c_ok && on_map && wgt != 0.f) {
        const int y = (int)((unsigned)rec.w >> 16) - 32768 + dy, x = (int)((unsigned)rec.w & 0xFFFFu) - 32768 + dxw;
        atomicAdd(xb + ((int64_t)y * G.W + x) * G.C + c, add);
      }
    }
  };
  // the column gradients of pixel px+1 are in flight while pixel px is scattered into the window
  float ga[16], gb[16];
  load_px(0, ga);
  for (int px = 0; px < COL2IM_T * COL2IM_T; px += 2) {
    load_px(px + 1, gb);
    scatter_px(px, ga);
    load_px(px + 2, ga);
    scatter_px(px + 1, gb);
  }
  __syncthreads();
  // flush: a wave adds whole cells (64 channels = 256 contiguous bytes per atomic instruction)
  for (int i = wave; i < cells; i += 4) {
    const int y = wy0 + i / win_w, x = wx0 + i % win_w;
    if ((unsigned)y >= (unsigned)G.H || (unsigned)x >= (unsigned)G.W) continue;
    const float v = win[i * COL2IM_CP + lane];
    if (__builtin_amdgcn_ballot_w64(v != 0.f) == 0) continue;             // an untouched cell: nothing to add
    if (c0 + lane < G.C) atomicAdd(xb + ((int64_t)y * G.W + x) * G.C + c0 + lane, v);
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
  // tiled LDS-window kernel (a sixth of the float-atomic traffic) when a 64-channel slab lies inside one deformable
  // group and the window fits in LDS; the per-sample scatter otherwise
  static const int tiled = [] { const char* v = getenv("CPM_DEFORM_TILED"); return v ? atoi(v) : 1; }();
  const int win_h = (COL2IM_T - 1) * stride + (R - 1) * dilation + 2 + 2 * COL2IM_MG;
  const int win_w = (COL2IM_T - 1) * stride + (S - 1) * dilation + 2 + 2 * COL2IM_MG;
  const size_t lds = (size_t)win_h * win_w * COL2IM_CP * sizeof(float);
  // (integer sampling positions -- offset == NULL, the narrow-group 3x3s -- touch one corner per sample: the scatter
  // kernel then issues 9 atomics per output pixel and is the faster one)
  if (tiled && offset && R * S <= 16 && (C / deformable_groups) % 64 == 0 && lds <= 64 * 1024) {
    const unsigned tiles = (unsigned)(N * cpm::cdiv(P, COL2IM_T) * cpm::cdiv(Q, COL2IM_T));
    hipLaunchKernelGGL(deform_col2im_tiled_kernel, dim3(tiles, (unsigned)cpm::cdiv(C, 64)), dim3(256), lds,
                       (hipStream_t)stream, dcols, offset, G, dx, win_h, win_w);
    return cpm::check_launch("deform_col2im (tiled)");
  }
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
