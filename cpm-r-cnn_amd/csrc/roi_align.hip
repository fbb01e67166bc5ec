// RoIAlign forward / backward for gfx950.
//
// Semantics follow the reference's RoIAlign exactly (same expression trees, built with
// -ffp-contract=off so no FMA is formed where the reference has a multiply and an add):
//   geometry       pet/lib/ops/csrc/ROIAlign/ROIAlign_cuda.cu:196-232 == ROIAlign_cpu.cpp:191-222
//   bilinear taps  ROIAlign_cuda.cu:12-63      nearest  :66-88
//   backward       ROIAlign_cuda.cu:259-365 (atomic scatter of g*w/count)
//   level mapping  pet/rcnn/utils/poolers.py:30-40 (fused variant)
//
// Design (not the reference's one-thread-per-output gather): the resident layout is NHWC, a
// workgroup owns one RoI x a chunk of output bins, each wave owns a bin at a time and its 64
// lanes sweep the channels with 16-byte loads.  The (y,x) taps and weights of a bin are
// wave-uniform, every load / store / atomic of a wave is one contiguous run of channels.
#include "common.h"

namespace {

struct Tap {
  int p0, p1, p2, p3;  // pixel indices y*W+x of the four corners
  float w0, w1, w2, w3;
  bool valid;
};

__device__ __forceinline__ Tap bilinear_tap(int H, int W, float y, float x) {
  Tap t;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    t.p0 = t.p1 = t.p2 = t.p3 = 0;
    t.w0 = t.w1 = t.w2 = t.w3 = 0.f;
    t.valid = false;
    return t;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y0 = (int)y, x0 = (int)x, y1, x1;
  if (y0 >= H - 1) { y1 = y0 = H - 1; y = (float)y0; } else { y1 = y0 + 1; }
  if (x0 >= W - 1) { x1 = x0 = W - 1; x = (float)x0; } else { x1 = x0 + 1; }
  float ly = y - (float)y0, lx = x - (float)x0;
  float hy = 1.f - ly, hx = 1.f - lx;
  t.p0 = y0 * W + x0; t.p1 = y0 * W + x1; t.p2 = y1 * W + x0; t.p3 = y1 * W + x1;
  t.w0 = hy * hx; t.w1 = hy * lx; t.w2 = ly * hx; t.w3 = ly * lx;
  t.valid = true;
  return t;
}

__device__ __forceinline__ int nearest_tap(int H, int W, float y, float x) {
  if (y < -0.5f || y >= (float)H - 0.5f || x < -0.5f || x >= (float)W - 0.5f) return -1;
  return (int)roundf(y) * W + (int)roundf(x);
}

struct Geom {
  float start_w, start_h, bin_w, bin_h;
  int grid_h, grid_w, batch;
};

__device__ __forceinline__ Geom roi_geometry(const float* __restrict__ roi, float scale, int PH, int PW,
                                             int sampling_ratio, bool aligned) {
  Geom g;
  g.batch = (int)roi[0];
  float off = aligned ? 0.5f : 0.0f;
  g.start_w = roi[1] * scale - off;
  g.start_h = roi[2] * scale - off;
  float end_w = roi[3] * scale - off;
  float end_h = roi[4] * scale - off;
  float rw = end_w - g.start_w, rh = end_h - g.start_h;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  g.bin_h = rh / (float)PH;
  g.bin_w = rw / (float)PW;
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
  return g;
}

struct Level { const float* in; float* gin; int H, W; float scale; };
struct Levels {
  Level l[5];
  int n;
  float k_min, k_max, s0, lvl0, eps;
};

// poolers.py:35-40 with BoxList.area's +1 (bounding_box.py:309-310)
__device__ __forceinline__ int map_level(const float* __restrict__ roi, const Levels& L) {
  float area = (roi[3] - roi[1] + 1.f) * (roi[4] - roi[2] + 1.f);
  float s = sqrtf(area);
  float lv = floorf(L.lvl0 + log2f(s / L.s0 + L.eps));
  lv = fminf(fmaxf(lv, L.k_min), L.k_max);
  return (int)lv - (int)L.k_min;
}

constexpr int BINS_PER_WAVE = 4;
constexpr int WAVES = 4;

// ---- NHWC forward -----------------------------------------------------------------------------------
// grid (K, ceil(PH*PW / (WAVES*BINS_PER_WAVE))), block 256.  FPN=true: level picked per RoI.
template <bool FPN, int INTERP>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const float* __restrict__ input, Levels L,
                                                          const float* __restrict__ rois, int B, int C, int H,
                                                          int W, float scale, int PH, int PW, int sampling_ratio,
                                                          bool aligned, float* __restrict__ out,
                                                          int32_t* __restrict__ levels_out) {
  const int n = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* roi = rois + 5 * (size_t)n;
  if (FPN) {
    int lv = map_level(roi, L);
    input = L.l[lv].in; H = L.l[lv].H; W = L.l[lv].W; scale = L.l[lv].scale;
    if (levels_out && blockIdx.y == 0 && threadIdx.x == 0) levels_out[n] = lv;
  }
  Geom g = roi_geometry(roi, scale, PH, PW, sampling_ratio, aligned);
  const int nbins = PH * PW;
  const bool batch_ok = g.batch >= 0 && g.batch < B;
  const float* src = input + (size_t)(batch_ok ? g.batch : 0) * H * W * C;
  const int ng = g.grid_h * g.grid_w;
  const float count = (float)(ng > 1 ? ng : 1);
  const int bin0 = (blockIdx.y * WAVES + wave) * BINS_PER_WAVE;
  const bool vec = (C & 3) == 0;
  for (int bi = 0; bi < BINS_PER_WAVE; ++bi) {
    const int bin = bin0 + bi;
    if (bin >= nbins) break;
    const int ph = bin / PW, pw = bin - ph * PW;
    float* dst = out + ((size_t)n * nbins + bin) * C;
    if (vec) {
      for (int c = lane * 4; c < C; c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int iy = 0; iy < g.grid_h; ++iy) {
          const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
          for (int ix = 0; ix < g.grid_w; ++ix) {
            const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
            if (INTERP == 0) {
              Tap t = bilinear_tap(H, W, yy, xx);
              if (!batch_ok) continue;
              // invalid taps have zero weights and index 0: the reference still evaluates them (adds +0)
              const float4 v0 = *(const float4*)(src + (size_t)t.p0 * C + c);
              const float4 v1 = *(const float4*)(src + (size_t)t.p1 * C + c);
              const float4 v2 = *(const float4*)(src + (size_t)t.p2 * C + c);
              const float4 v3 = *(const float4*)(src + (size_t)t.p3 * C + c);
              acc.x += t.w0 * v0.x + t.w1 * v1.x + t.w2 * v2.x + t.w3 * v3.x;
              acc.y += t.w0 * v0.y + t.w1 * v1.y + t.w2 * v2.y + t.w3 * v3.y;
              acc.z += t.w0 * v0.z + t.w1 * v1.z + t.w2 * v2.z + t.w3 * v3.z;
              acc.w += t.w0 * v0.w + t.w1 * v1.w + t.w2 * v2.w + t.w3 * v3.w;
            } else {
              int p = nearest_tap(H, W, yy, xx);
              if (p >= 0 && batch_ok) {
                const float4 v = *(const float4*)(src + (size_t)p * C + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
              }
            }
          }
        }
        acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
        *(float4*)(dst + c) = acc;
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        float acc = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
          const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
          for (int ix = 0; ix < g.grid_w; ++ix) {
            const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
            if (INTERP == 0) {
              Tap t = bilinear_tap(H, W, yy, xx);
              if (!batch_ok) continue;
              acc += t.w0 * src[(size_t)t.p0 * C + c] + t.w1 * src[(size_t)t.p1 * C + c] +
                     t.w2 * src[(size_t)t.p2 * C + c] + t.w3 * src[(size_t)t.p3 * C + c];
            } else {
              int p = nearest_tap(H, W, yy, xx);
              if (p >= 0 && batch_ok) acc += src[(size_t)p * C + c];
            }
          }
        }
        dst[c] = acc / count;
      }
    }
  }
}

// ---- NHWC backward: wave-contiguous float atomics (256 B per wave-instruction) ------------------------
template <bool FPN, int INTERP>
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc(const float* __restrict__ grad, Levels L,
                                                          const float* __restrict__ rois, int B, int C, int H,
                                                          int W, float scale, int PH, int PW, int sampling_ratio,
                                                          bool aligned, float* __restrict__ gin) {
  const int n = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float* roi = rois + 5 * (size_t)n;
  if (FPN) {
    int lv = map_level(roi, L);
    gin = L.l[lv].gin; H = L.l[lv].H; W = L.l[lv].W; scale = L.l[lv].scale;
  }
  Geom g = roi_geometry(roi, scale, PH, PW, sampling_ratio, aligned);
  if (g.batch < 0 || g.batch >= B) return;
  const int nbins = PH * PW;
  float* dst = gin + (size_t)g.batch * H * W * C;
  const float count = (float)(g.grid_h * g.grid_w);
  const int bin0 = (blockIdx.y * WAVES + wave) * BINS_PER_WAVE;
  for (int bi = 0; bi < BINS_PER_WAVE; ++bi) {
    const int bin = bin0 + bi;
    if (bin >= nbins) break;
    const int ph = bin / PW, pw = bin - ph * PW;
    const float* gsrc = grad + ((size_t)n * nbins + bin) * C;
    if (INTERP == 0 && g.grid_h * g.grid_w <= 4) {
      // The kernel runs at the chip's float-atomic rate, and the 2x2 samples of a bin sit half a bin apart: on the
      // RoI's own pyramid level a 7x7 bin spans 1-2 pixels and a 14x14 bin half of that, so the samples' corner
      // pixels overlap (typically 9 -- or 4 -- distinct pixels instead of 16).  All of this is wave-uniform:
      // collect the (pixel, weight) pairs of the bin, merge equal pixels, drop zero weights, then one atomic per
      // distinct pixel and channel.
      int px[16];
      float wt[16];
      int ne = 0;
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
          const Tap t = bilinear_tap(H, W, yy, xx);
          if (!t.valid) continue;
          const int tp[4] = {t.p0, t.p1, t.p2, t.p3};
          const float tw[4] = {t.w0 / count, t.w1 / count, t.w2 / count, t.w3 / count};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (tw[k] == 0.f) continue;
            bool merged = false;
            for (int j = 0; j < ne; ++j)
              if (px[j] == tp[k]) { wt[j] += tw[k]; merged = true; break; }
            if (!merged) { px[ne] = tp[k]; wt[ne] = tw[k]; ++ne; }
          }
        }
      }
      for (int c = lane; c < C; c += 64) {
        const float go = gsrc[c];
        for (int j = 0; j < ne; ++j) atomicAdd(dst + (size_t)px[j] * C + c, go * wt[j]);
      }
      continue;
    }
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
        if (INTERP == 0) {
          Tap t = bilinear_tap(H, W, yy, xx);
          if (!t.valid) continue;
          for (int c = lane; c < C; c += 64) {
            const float go = gsrc[c];
            atomicAdd(dst + (size_t)t.p0 * C + c, go * t.w0 / count);
            atomicAdd(dst + (size_t)t.p1 * C + c, go * t.w1 / count);
            atomicAdd(dst + (size_t)t.p2 * C + c, go * t.w2 / count);
            atomicAdd(dst + (size_t)t.p3 * C + c, go * t.w3 / count);
          }
        } else {
          int p = nearest_tap(H, W, yy, xx);
          if (p < 0) continue;
          for (int c = lane; c < C; c += 64) atomicAdd(dst + (size_t)p * C + c, gsrc[c] / count);
        }
      }
    }
  }
}

// ---- NCHW (the reference's layout): one thread per output element, grid-stride ------------------------
template <int INTERP>
__global__ __launch_bounds__(256) void roi_align_fwd_nchw(const float* __restrict__ input,
                                                          const float* __restrict__ rois, int64_t total, int B,
                                                          int C, int H, int W, float scale, int PH, int PW,
                                                          int sampling_ratio, bool aligned,
                                                          float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int pw = idx % PW, ph = (idx / PW) % PH, c = (idx / PW / PH) % C, n = idx / PW / PH / C;
    Geom g = roi_geometry(rois + 5 * (size_t)n, scale, PH, PW, sampling_ratio, aligned);
    if (g.batch < 0 || g.batch >= B) { out[idx] = 0.f; continue; }
    const float* src = input + ((size_t)g.batch * C + c) * H * W;
    const int ng = g.grid_h * g.grid_w;
    const float count = (float)(ng > 1 ? ng : 1);
    float acc = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
        if (INTERP == 0) {
          Tap t = bilinear_tap(H, W, yy, xx);
          acc += t.w0 * src[t.p0] + t.w1 * src[t.p1] + t.w2 * src[t.p2] + t.w3 * src[t.p3];
        } else {
          int p = nearest_tap(H, W, yy, xx);
          if (p >= 0) acc += src[p];
        }
      }
    }
    out[idx] = acc / count;
  }
}

template <int INTERP>
__global__ __launch_bounds__(256) void roi_align_bwd_nchw(const float* __restrict__ grad,
                                                          const float* __restrict__ rois, int64_t total, int B,
                                                          int C, int H, int W, float scale, int PH, int PW,
                                                          int sampling_ratio, bool aligned,
                                                          float* __restrict__ gin) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int pw = idx % PW, ph = (idx / PW) % PH, c = (idx / PW / PH) % C, n = idx / PW / PH / C;
    Geom g = roi_geometry(rois + 5 * (size_t)n, scale, PH, PW, sampling_ratio, aligned);
    if (g.batch < 0 || g.batch >= B) continue;
    float* dst = gin + ((size_t)g.batch * C + c) * H * W;
    const float count = (float)(g.grid_h * g.grid_w);
    const float go = grad[idx];
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
        if (INTERP == 0) {
          Tap t = bilinear_tap(H, W, yy, xx);
          if (!t.valid) continue;
          atomicAdd(dst + t.p0, go * t.w0 / count);
          atomicAdd(dst + t.p1, go * t.w1 / count);
          atomicAdd(dst + t.p2, go * t.w2 / count);
          atomicAdd(dst + t.p3, go * t.w3 / count);
        } else {
          int p = nearest_tap(H, W, yy, xx);
          if (p >= 0) atomicAdd(dst + p, go / count);
        }
      }
    }
  }
}

// ---- PoolPointsInterp (NCHW, reference layout): PoolPointsInterp_cuda.cu:62-91,147-196 ----------------
__global__ void pool_points_fwd(const float* __restrict__ input, const float* __restrict__ pts, int64_t total,
                                int B, int C, int H, int W, float scale, float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = idx % C, n = idx / C;
    const int b = n / 196;
    const float X = pts[3 * (size_t)n + 1] * scale, Y = pts[3 * (size_t)n + 2] * scale;
    float v = 0.f;
    if (b < B) {
      Tap t = bilinear_tap(H, W, Y, X);
      const float* src = input + ((size_t)b * C + c) * H * W;
      if (t.valid) v = t.w0 * src[t.p0] + t.w1 * src[t.p1] + t.w2 * src[t.p2] + t.w3 * src[t.p3];
    }
    out[idx] = v;
  }
}

__global__ void pool_points_bwd(const float* __restrict__ grad, const float* __restrict__ pts, int64_t total,
                                int B, int C, int H, int W, float scale, float* __restrict__ gin) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = idx % C, n = idx / C;
    const int b = n / 196;
    if (b >= B) continue;
    const float X = pts[3 * (size_t)n + 1] * scale, Y = pts[3 * (size_t)n + 2] * scale;
    Tap t = bilinear_tap(H, W, Y, X);
    if (!t.valid) continue;
    float* dst = gin + ((size_t)b * C + c) * H * W;
    const float go = grad[idx];
    atomicAdd(dst + t.p0, go * t.w0);
    atomicAdd(dst + t.p1, go * t.w1);
    atomicAdd(dst + t.p2, go * t.w2);
    atomicAdd(dst + t.p3, go * t.w3);
  }
}

int grid_1d(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

CPM_EXPORT int cpm_roi_align_forward(const float* input, const float* rois, int K, int B, int C, int H, int W,
                                     float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                     int aligned, int interp, int layout, float* output, void* stream) {
  CPM_REQUIRE(interp == 0 || interp == 1, "interpolation must be bilinear (0) or nearest (1)");
  CPM_REQUIRE(layout == CPM_LAYOUT_NCHW || layout == CPM_LAYOUT_NHWC, "bad layout");
  CPM_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0, "bad shape");
  if (K == 0) return CPM_OK;
  CPM_REQUIRE(input && rois && output, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (layout == CPM_LAYOUT_NHWC) {
    Levels L = {};
    dim3 grid(K, cpm::cdiv(pooled_h * pooled_w, WAVES * BINS_PER_WAVE));
    if (interp == 0)
      hipLaunchKernelGGL((roi_align_fwd_nhwc<false, 0>), grid, dim3(256), 0, s, input, L, rois, B, C, H, W,
                         spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, output, nullptr);
    else
      hipLaunchKernelGGL((roi_align_fwd_nhwc<false, 1>), grid, dim3(256), 0, s, input, L, rois, B, C, H, W,
                         spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, output, nullptr);
  } else {
    int64_t total = (int64_t)K * C * pooled_h * pooled_w;
    if (interp == 0)
      hipLaunchKernelGGL((roi_align_fwd_nchw<0>), dim3(grid_1d(total)), dim3(256), 0, s, input, rois, total, B, C,
                         H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, output);
    else
      hipLaunchKernelGGL((roi_align_fwd_nchw<1>), dim3(grid_1d(total)), dim3(256), 0, s, input, rois, total, B, C,
                         H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, output);
  }
  return cpm::check_launch("roi_align_forward");
}

CPM_EXPORT int cpm_roi_align_backward(const float* grad_output, const float* rois, int K, int B, int C, int H,
                                      int W, float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                      int aligned, int interp, int layout, float* grad_input, void* stream) {
  CPM_REQUIRE(interp == 0 || interp == 1, "interpolation must be bilinear (0) or nearest (1)");
  CPM_REQUIRE(layout == CPM_LAYOUT_NCHW || layout == CPM_LAYOUT_NHWC, "bad layout");
  CPM_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0, "bad shape");
  if (K == 0) return CPM_OK;
  CPM_REQUIRE(grad_output && rois && grad_input, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (layout == CPM_LAYOUT_NHWC) {
    Levels L = {};
    dim3 grid(K, cpm::cdiv(pooled_h * pooled_w, WAVES * BINS_PER_WAVE));
    if (interp == 0)
      hipLaunchKernelGGL((roi_align_bwd_nhwc<false, 0>), grid, dim3(256), 0, s, grad_output, L, rois, B, C, H, W,
                         spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, grad_input);
    else
      hipLaunchKernelGGL((roi_align_bwd_nhwc<false, 1>), grid, dim3(256), 0, s, grad_output, L, rois, B, C, H, W,
                         spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, grad_input);
  } else {
    int64_t total = (int64_t)K * C * pooled_h * pooled_w;
    if (interp == 0)
      hipLaunchKernelGGL((roi_align_bwd_nchw<0>), dim3(grid_1d(total)), dim3(256), 0, s, grad_output, rois, total, B,
                         C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, grad_input);
    else
      hipLaunchKernelGGL((roi_align_bwd_nchw<1>), dim3(grid_1d(total)), dim3(256), 0, s, grad_output, rois, total, B,
                         C, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned != 0, grad_input);
  }
  return cpm::check_launch("roi_align_backward");
}

static int fill_levels(Levels& L, const float* const* feats, float* const* gfeats, const int* hs, const int* ws,
                       const float* scales, int num_levels, float k_min, float k_max, float s0, float lvl0,
                       float eps) {
  if (num_levels < 1 || num_levels > 5) return CPM_EINVAL;
  if ((int)(k_max - k_min) + 1 > num_levels) return CPM_EINVAL;
  L.n = num_levels; L.k_min = k_min; L.k_max = k_max; L.s0 = s0; L.lvl0 = lvl0; L.eps = eps;
  for (int i = 0; i < num_levels; ++i) {
    L.l[i].in = feats ? feats[i] : nullptr;
    L.l[i].gin = gfeats ? gfeats[i] : nullptr;
    L.l[i].H = hs[i]; L.l[i].W = ws[i]; L.l[i].scale = scales[i];
    if (hs[i] <= 0 || ws[i] <= 0) return CPM_EINVAL;
  }
  return CPM_OK;
}

CPM_EXPORT int cpm_roi_align_fpn_forward(const float* const* feats, const int* hs, const int* ws,
                                         const float* scales, int num_levels, const float* rois, int K, int B,
                                         int C, int pooled_h, int pooled_w, int sampling_ratio, float k_min,
                                         float k_max, float canonical_scale, float canonical_level, float eps,
                                         float* output, int32_t* levels_out, void* stream) {
  CPM_REQUIRE(K >= 0 && B > 0 && C > 0 && pooled_h > 0 && pooled_w > 0, "bad shape");
  if (K == 0) return CPM_OK;
  CPM_REQUIRE(feats && hs && ws && scales && rois && output, "null pointer");
  Levels L = {};
  CPM_REQUIRE(fill_levels(L, feats, nullptr, hs, ws, scales, num_levels, k_min, k_max, canonical_scale,
                          canonical_level, eps) == CPM_OK, "bad level table");
  dim3 grid(K, cpm::cdiv(pooled_h * pooled_w, WAVES * BINS_PER_WAVE));
  hipLaunchKernelGGL((roi_align_fwd_nhwc<true, 0>), grid, dim3(256), 0, (hipStream_t)stream, nullptr, L, rois, B, C,
                     0, 0, 0.f, pooled_h, pooled_w, sampling_ratio, false, output, levels_out);
  return cpm::check_launch("roi_align_fpn_forward");
}

CPM_EXPORT int cpm_roi_align_fpn_backward(const float* grad_output, float* const* grad_feats, const int* hs,
                                          const int* ws, const float* scales, int num_levels, const float* rois,
                                          int K, int B, int C, int pooled_h, int pooled_w, int sampling_ratio,
                                          float k_min, float k_max, float canonical_scale, float canonical_level,
                                          float eps, void* stream) {
  CPM_REQUIRE(K >= 0 && B > 0 && C > 0 && pooled_h > 0 && pooled_w > 0, "bad shape");
  if (K == 0) return CPM_OK;
  CPM_REQUIRE(grad_output && grad_feats && hs && ws && scales && rois, "null pointer");
  Levels L = {};
  CPM_REQUIRE(fill_levels(L, nullptr, grad_feats, hs, ws, scales, num_levels, k_min, k_max, canonical_scale,
                          canonical_level, eps) == CPM_OK, "bad level table");
  dim3 grid(K, cpm::cdiv(pooled_h * pooled_w, WAVES * BINS_PER_WAVE));
  hipLaunchKernelGGL((roi_align_bwd_nhwc<true, 0>), grid, dim3(256), 0, (hipStream_t)stream, grad_output, L, rois, B,
                     C, 0, 0, 0.f, pooled_h, pooled_w, sampling_ratio, false, nullptr);
  return cpm::check_launch("roi_align_fpn_backward");
}

CPM_EXPORT int cpm_pool_points_interp_forward(const float* input, const float* pts, int K, int B, int C, int H,
                                              int W, float spatial_scale, float* output, void* stream) {
  CPM_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0, "bad shape");
  if (K == 0) return CPM_OK;
  CPM_REQUIRE(input && pts && output, "null pointer");
  int64_t total = (int64_t)K * C;
  hipLaunchKernelGGL(pool_points_fwd, dim3(grid_1d(total)), dim3(256), 0, (hipStream_t)stream, input, pts, total, B,
                     C, H, W, spatial_scale, output);
  return cpm::check_launch("pool_points_interp_forward");
}

CPM_EXPORT int cpm_pool_points_interp_backward(const float* grad_output, const float* pts, int K, int B, int C,
                                               int H, int W, float spatial_scale, float* grad_input,
                                               void* stream) {
  CPM_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0, "bad shape");
  if (K == 0) return CPM_OK;
  CPM_REQUIRE(grad_output && pts && grad_input, "null pointer");
  int64_t total = (int64_t)K * C;
  hipLaunchKernelGGL(pool_points_bwd, dim3(grid_1d(total)), dim3(256), 0, (hipStream_t)stream, grad_output, pts,
                     total, B, C, H, W, spatial_scale, grad_input);
  return cpm::check_launch("pool_points_interp_backward");
}
