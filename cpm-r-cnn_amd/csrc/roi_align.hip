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

// ---- NHWC multi-level backward as a GATHER (no float atomics) ------------------------------------------
// The scatter kernel above runs at the chip's float-atomic rate: K*C*(4..16)*PH*PW atomics, many of them onto
// pixels that other bins and other RoIs also hit.  Here every 8x8 tile of every pyramid level is owned by one
// workgroup: a binning pass lists the RoIs whose footprint touches the tile, the workgroup turns each listed RoI
// into two small dense tables -- WY[tile row][ph] and WX[tile col][pw], the summed bilinear weights (incl. the
// reference's clamping rules) of the RoI's sample rows / columns on that pixel row / column -- and a pixel's
// gradient is  sum_roi sum_{ph,pw} WY[r][ph]*WX[c][pw]/count * g[roi,ph,pw,:]  accumulated in registers with
// 1 KB wave-contiguous loads of g and written ONCE (no atomics; the per-tile RoI list is sorted, so the result is
// bit-reproducible).  Only tiles that some RoI reaches are visited (persistent workgroups over a compacted list);
// a fresh map is cleared by a memset first.  Measured (K=1024, 7x7, 2 x 256-ch pyramid): 245 us + ~50 us of fills vs
// 480 + 50 for the scatter kernel.
constexpr int GT = 8;                 // tile edge (64 pixels: 4 per wavefront, 16 wavefronts)
constexpr int GCHUNK = 32;            // RoIs staged per pass
constexpr int GBINS = 16;             // max pooled size per dimension
constexpr int GLIST = 128;            // (weight, row) pairs collected per wave before they are streamed
constexpr int GTHREADS = 1024;
constexpr int GPPW = GT * GT / (GTHREADS / 64);   // pixels per wavefront

struct GatherLevels {
  int tile_base[6];                   // first tile id of each level (all batches), [n] = total
  int tiles_y[5], tiles_x[5];
};

// Several RoI sets differentiated by ONE pass over the tiles: the heads of a training step pool the same pyramid with
// their own RoIs, pooled sizes and gradient tensors (cls 7x7, three grid stages 14x14, RSM 7x7).  RoI n of the launch
// belongs to set s with first[s] <= n < first[s + 1]; its gradient rows are grad[s] + ((n - first[s]) * ph * pw + bin) * C.
constexpr int GSETS = 8;
struct RoiSets {
  const float* rois[GSETS];
  const float* grad[GSETS];
  int first[GSETS + 1];
  int ph[GSETS], pw[GSETS], ratio[GSETS];
  int n;
};

__device__ __forceinline__ int set_of(const RoiSets& S, int n) {
  int s = 0;
  while (s + 1 < S.n && n >= S.first[s + 1]) ++s;
  return s;
}

__global__ void __launch_bounds__(64) roi_bin_tiles(Levels L, GatherLevels G, RoiSets S, int K,
                                                    int B, int cap, int* __restrict__ tile_count,
                                                    int* __restrict__ tile_list) {
  const int n = blockIdx.x;                                  // one wavefront per RoI, lanes sweep its tiles
  const int set = set_of(S, n);
  const int PH = S.ph[set], PW = S.pw[set];
  const float* roi = S.rois[set] + 5 * (size_t)(n - S.first[set]);
  const int lv = map_level(roi, L);
  const int H = L.l[lv].H, W = L.l[lv].W;
  const Geom g = roi_geometry(roi, L.l[lv].scale, PH, PW, 1, false);
  if (g.batch < 0 || g.batch >= B) return;
  // conservative pixel box of all bilinear corners: samples lie in [start, start + PH*bin]
  int y0 = (int)floorf(g.start_h) - 1, y1 = (int)floorf(g.start_h + (float)PH * g.bin_h) + 2;
  int x0 = (int)floorf(g.start_w) - 1, x1 = (int)floorf(g.start_w + (float)PW * g.bin_w) + 2;
  y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, H - 1); x1 = min(x1, W - 1);
  if (y1 < y0 || x1 < x0) return;
  const int base = G.tile_base[lv] + g.batch * G.tiles_y[lv] * G.tiles_x[lv];
  const int ty0 = y0 / GT, tx0 = x0 / GT, nty = y1 / GT - ty0 + 1, ntx = x1 / GT - tx0 + 1;
  for (int i = threadIdx.x; i < nty * ntx; i += 64) {
    const int t = base + (ty0 + i / ntx) * G.tiles_x[lv] + tx0 + i % ntx;
    const int slot = atomicAdd(&tile_count[t], 1);
    if (slot < cap) tile_list[(size_t)t * cap + slot] = n;
  }
}

// tiles with at least one RoI, in any order (most tiles of the fine levels are empty: the gather kernel only visits
// these, the rest of a fresh map is cleared by a plain memset)
__global__ void __launch_bounds__(256) roi_active_tiles(const int* __restrict__ tile_count, int tiles,
                                                        int* __restrict__ active_count, int* __restrict__ active) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < tiles && tile_count[t] > 0) active[atomicAdd(active_count, 1)] = t;
}

// Summed bilinear weights of one RoI's sample rows (or columns) on ONE pixel row (column) `pix` of a tile:
// row[p] for the bins p in [first, first + cnt).  Only the samples that can reach the pixel are evaluated: those
// whose coordinate lies within one pixel of it (the reference's clamping keeps that true at both borders); the index
// range is found by inverting the sample spacing with a margin of one, the coordinate itself is the forward's
// expression.
__device__ void axis_row(float start, float bin, int grid, int P, int size, int pix, float* __restrict__ row,
                         int* __restrict__ first, int* __restrict__ cnt) {
  const float sp = bin / (float)grid;
  const int total = P * grid;
  float flo = floorf(((float)(pix - 1) - start) / sp - 0.5f) - 1.f;
  float fhi = ceilf(((float)(pix + 1) - start) / sp - 0.5f) + 1.f;
  if (!(flo > -1.f)) flo = 0.f;                              // also catches NaN
  if (!(fhi < (float)total)) fhi = (float)(total - 1);
  const int s_lo = (int)flo, s_hi = (int)fhi;
  if (s_hi < s_lo) { *first = 0; *cnt = 0; return; }
  const int p_lo = s_lo / grid, p_hi = s_hi / grid;
  // the samples arrive in bin order: a bin's sum is formed in a register (0 + the terms in sample order, as a
  // read-modify-write of row[p] per term would) and stored once
  int p = p_lo, i = s_lo - p_lo * grid;
  int nz_lo = 1 << 20, nz_hi = -1;                           // the bins whose sum is not zero (a contiguous run)
  float sum = 0.f;
  for (int sidx = s_lo; sidx <= s_hi; ++sidx) {
    float y = start + (float)p * bin + ((float)i + .5f) * bin / (float)grid;       // the forward's expression
    if (!(y < -1.0f || y > (float)size)) {
      if (y <= 0.f) y = 0.f;
      int lo = (int)y, hi;
      if (lo >= size - 1) { hi = lo = size - 1; y = (float)lo; } else { hi = lo + 1; }
      const float l = y - (float)lo, h = 1.f - l;
      if (lo == pix) sum += h;
      if (hi == pix) sum += l;
    }
    if (++i == grid) {
      row[p] = sum;
      if (sum != 0.f) { nz_lo = min(nz_lo, p); nz_hi = p; }
      sum = 0.f;
      i = 0;
      ++p;
    }
  }
  if (i != 0) {
    row[p] = sum;
    if (sum != 0.f) { nz_lo = min(nz_lo, p); nz_hi = p; }
  }
  // (the index range above carries a margin of a sample on either side: the bins it adds weigh exactly zero and
  // contribute nothing; reporting the non-zero run keeps them out of the consumers' loops and staging)
  *first = nz_hi >= nz_lo ? nz_lo : 0;
  *cnt = nz_hi >= nz_lo ? nz_hi - nz_lo + 1 : 0;
}

__global__ void __launch_bounds__(GTHREADS) roi_align_bwd_gather(Levels L, GatherLevels G, RoiSets S, int C, int cap,
                                                            const int* __restrict__ tile_count,
                                                            const int* __restrict__ tile_list,
                                                            const int* __restrict__ active_count,
                                                            const int* __restrict__ active, int accumulate_mask) {
  __shared__ const float* s_grad[GSETS];                    // the sets' gradient tensors, indexed per list entry
  __shared__ int s_pw[GCHUNK];                              // pooled width of the chunk's RoIs (bin index = row * pw + col)
  if (threadIdx.x < GSETS) s_grad[threadIdx.x] = threadIdx.x < S.n ? S.grad[threadIdx.x] : nullptr;
  __shared__ float s_wy[GCHUNK][GT * GBINS + 1];            // +1: lane-per-RoI reads hit distinct banks
  __shared__ float s_wx[GCHUNK][GT * GBINS + 1];
  __shared__ int s_fy[GCHUNK][GT + 1], s_ny[GCHUNK][GT + 1], s_fx[GCHUNK][GT + 1], s_nx[GCHUNK][GT + 1];
  __shared__ int s_roi[GCHUNK];
  __shared__ float s_inv[GCHUNK];
  __shared__ int s_sorted[1024];
  __shared__ float s_ew[GTHREADS / 64][GLIST];
  __shared__ int s_eo[GTHREADS / 64][GLIST];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n_active = *active_count;
  for (int ai = blockIdx.x; ai < n_active; ai += gridDim.x) {          // persistent over the non-empty tiles
  __syncthreads();                                                      // LDS of the previous tile is free
  const int tile = active[ai];
  int lv = 0;
  while (lv + 1 < L.n && tile >= G.tile_base[lv + 1]) ++lv;
  const int H = L.l[lv].H, W = L.l[lv].W;
  const int per_img = G.tiles_y[lv] * G.tiles_x[lv];
  const int rel = tile - G.tile_base[lv];
  const int b = rel / per_img, tt = rel - b * per_img;
  const int ty0 = (tt / G.tiles_x[lv]) * GT, tx0 = (tt % G.tiles_x[lv]) * GT;
  const bool acc_level = (accumulate_mask >> lv) & 1;
  float* dst = L.l[lv].gin + (size_t)b * H * W * C;
  int nlist = tile_count[tile];
  if (nlist > cap) nlist = cap;
  const int* list = tile_list + (size_t)tile * cap;
  // deterministic summation order: sort the (short) list by RoI index
  const bool sorted = nlist <= 1024;
  if (sorted && nlist > 0) {
    int np2 = 1;
    while (np2 < nlist) np2 <<= 1;
    for (int i = tid; i < np2; i += GTHREADS) s_sorted[i] = i < nlist ? list[i] : 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < np2 / 2; t += GTHREADS) {
          const int lo = ((t / stride) * stride * 2) + (t % stride), hi = lo + stride;
          const bool up = (lo & size) == 0;
          const int a = s_sorted[lo], c2 = s_sorted[hi];
          if ((a > c2) == up) { s_sorted[lo] = c2; s_sorted[hi] = a; }
        }
        __syncthreads();
      }
  }
  for (int c0 = 0; c0 < C; c0 += 256) {                     // 64 lanes x float4 per pass over the channels
    const int c = c0 + lane * 4;
    const bool c_ok = c < C;
    float4 acc[GPPW];                                         // this wave's pixels
#pragma unroll
    for (int i = 0; i < GPPW; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int chunk = 0; chunk < nlist; chunk += GCHUNK) {
      const int nch = min(GCHUNK, nlist - chunk);
      __syncthreads();
      for (int wi = tid; wi < nch * 2 * GT; wi += GTHREADS) { // one work item per (RoI, axis, tile row / column)
        const int j = wi / (2 * GT), rem = wi - j * (2 * GT), axis = rem / GT, r = rem - axis * GT;
        const int n = sorted ? s_sorted[chunk + j] : list[chunk + j];
        const int set = set_of(S, n), PH = S.ph[set], PW = S.pw[set];
        const Geom g = roi_geometry(S.rois[set] + 5 * (size_t)(n - S.first[set]), L.l[lv].scale, PH, PW, S.ratio[set],
                                    false);
        if (axis == 0) {
          axis_row(g.start_h, g.bin_h, g.grid_h, PH, H, ty0 + r, &s_wy[j][r * GBINS], &s_fy[j][r], &s_ny[j][r]);
          if (r == 0) {
            // the RoI's first gradient row, tagged with its set (rows of one set stay below 2^24: K <= 8192, 256 bins)
            s_roi[j] = (set << 24) | ((n - S.first[set]) * PH * PW);
            s_pw[j] = PW;
            s_inv[j] = 1.f / (float)(g.grid_h * g.grid_w);
          }
        } else {
          axis_row(g.start_w, g.bin_w, g.grid_w, PW, W, tx0 + r, &s_wx[j][r * GBINS], &s_fx[j][r], &s_nx[j][r]);
        }
      }
      __syncthreads();
      // Per pixel: first collect the (weight, offset of the g row) pairs of every bin that reaches it -- wave-uniform
      // work -- into a small per-wave LDS list, then stream the list eight loads at a time.  With the loads issued in
      // the collecting loop each would wait out its full L2 latency behind a data-dependent branch.
#pragma unroll
      for (int pi = 0; pi < GPPW; ++pi) {
        const int pix = wave * GPPW + pi, r = pix / GT, cc = pix % GT;
        int ne = 0;
        auto flush = [&]() {
          if (c_ok) {
            int e = 0;
            for (; e + 8 <= ne; e += 8) {
              float4 v[8];
              float w8[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                w8[u] = s_ew[wave][e + u];
                const int eo = s_eo[wave][e + u];
                v[u] = *(const float4*)(s_grad[eo >> 24] + (size_t)(eo & 0xFFFFFF) * C + c);
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                acc[pi].x += w8[u] * v[u].x; acc[pi].y += w8[u] * v[u].y;
                acc[pi].z += w8[u] * v[u].z; acc[pi].w += w8[u] * v[u].w;
              }
            }
            for (; e < ne; ++e) {
              const float w = s_ew[wave][e];
              const int eo = s_eo[wave][e];
              const float4 v = *(const float4*)(s_grad[eo >> 24] + (size_t)(eo & 0xFFFFFF) * C + c);
              acc[pi].x += w * v.x; acc[pi].y += w * v.y; acc[pi].z += w * v.z; acc[pi].w += w * v.w;
            }
          }
          ne = 0;
        };
        // collection, one lane per RoI of the chunk: count this pixel's non-zero bins, exclusive prefix over the
        // lanes (RoI order = list order, so the summation order stays fixed), then each lane writes its entries
        int ny = 0, nx = 0, fy = 0, fx = 0, base = 0, cnt = 0, PW = 0;
        float inv = 0.f;
        if (lane < nch) {
          ny = s_ny[lane][r]; nx = s_nx[lane][cc];
          if (ny != 0 && nx != 0) {
            fy = s_fy[lane][r]; fx = s_fx[lane][cc];
            base = s_roi[lane];
            PW = s_pw[lane];
            inv = s_inv[lane];
            for (int a = 0; a < ny; ++a) {
              const float wy = s_wy[lane][r * GBINS + fy + a] * inv;
              if (wy == 0.f) continue;
              for (int q = 0; q < nx; ++q) cnt += (wy * s_wx[lane][cc * GBINS + fx + q] != 0.f) ? 1 : 0;
            }
          }
        }
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d, 64);
          if (lane >= d) incl += t;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total <= GLIST) {
          int slot = incl - cnt;
          if (cnt != 0) {
            for (int a = 0; a < ny; ++a) {
              const float wy = s_wy[lane][r * GBINS + fy + a] * inv;
              if (wy == 0.f) continue;
              for (int q = 0; q < nx; ++q) {
                const float w = wy * s_wx[lane][cc * GBINS + fx + q];
                if (w == 0.f) continue;
                s_ew[wave][slot] = w;
                s_eo[wave][slot] = base + (fy + a) * PW + fx + q;
                ++slot;
              }
            }
          }
          ne = total;
        } else {
          // more bins than the list holds (tiny RoIs spread over few pixels): wave-uniform walk with flushes
          for (int j = 0; j < nch; ++j) {
            const int jy = s_ny[j][r], jx = s_nx[j][cc];
            if (jy == 0 || jx == 0) continue;
            const int gy = s_fy[j][r], gx = s_fx[j][cc];
            const int jbase = s_roi[j], PW = s_pw[j];
            const float jinv = s_inv[j];
            for (int a = 0; a < jy; ++a) {
              const float wy = s_wy[j][r * GBINS + gy + a] * jinv;
              if (wy == 0.f) continue;
              for (int q = 0; q < jx; ++q) {
                const float w = wy * s_wx[j][cc * GBINS + gx + q];
                if (w == 0.f) continue;
                if (lane == 0) {
                  s_ew[wave][ne] = w;
                  s_eo[wave][ne] = jbase + (gy + a) * PW + gx + q;
                }
                if (++ne == GLIST) flush();
              }
            }
          }
        }
        flush();
      }
    }
    if (c_ok) {
#pragma unroll
      for (int pi = 0; pi < GPPW; ++pi) {
        const int pix = wave * GPPW + pi, y = ty0 + pix / GT, x = tx0 + pix % GT;
        if (y >= H || x >= W) continue;
        float4* o = (float4*)(dst + ((size_t)y * W + x) * C + c);
        if (acc_level) {
          float4 old = *o;
          old.x += acc[pi].x; old.y += acc[pi].y; old.z += acc[pi].z; old.w += acc[pi].w;
          *o = old;
        } else {
          *o = acc[pi];
        }
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

static void gather_levels(const Levels& L, int B, GatherLevels& G) {
  int t = 0;
  for (int i = 0; i < L.n; ++i) {
    G.tile_base[i] = t;
    G.tiles_y[i] = cpm::cdiv(L.l[i].H, GT);
    G.tiles_x[i] = cpm::cdiv(L.l[i].W, GT);
    t += B * G.tiles_y[i] * G.tiles_x[i];
  }
  for (int i = L.n; i <= 5; ++i) G.tile_base[i] = t;
}

CPM_EXPORT size_t cpm_roi_align_fpn_gather_workspace_bytes(const int* hs, const int* ws, int num_levels, int B, int K) {
  if (!hs || !ws || num_levels < 1 || num_levels > 5 || B <= 0 || K < 0) return 0;
  size_t tiles = 0;
  for (int i = 0; i < num_levels; ++i) tiles += (size_t)B * cpm::cdiv(hs[i], GT) * cpm::cdiv(ws[i], GT);
  return tiles * sizeof(int) * (2 + (size_t)(K > 0 ? K : 1)) + 64;
}

static int backward_gather_sets(const RoiSets& S, int K, float* const* grad_feats, const int* hs, const int* ws,
                                const float* scales, int num_levels, int B, int C, float k_min, float k_max,
                                float canonical_scale, float canonical_level, float eps, int accumulate_mask,
                                void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(K >= 0 && B > 0 && C > 0, "bad shape");
  CPM_REQUIRE(C % 4 == 0, "C must be a multiple of 4");
  CPM_REQUIRE(K <= 8192, "more than 8192 RoIs: use cpm_roi_align_fpn_backward");
  CPM_REQUIRE(grad_feats && hs && ws && scales, "null pointer");
  Levels L = {};
  CPM_REQUIRE(fill_levels(L, nullptr, grad_feats, hs, ws, scales, num_levels, k_min, k_max, canonical_scale,
                          canonical_level, eps) == CPM_OK, "bad level table");
  for (int i = 0; i < num_levels; ++i)
    CPM_REQUIRE(((uintptr_t)grad_feats[i] & 15) == 0, "gradient maps must be 16-byte aligned");
  GatherLevels G = {};
  gather_levels(L, B, G);
  const int tiles = G.tile_base[num_levels];
  const int cap = K > 0 ? K : 1;
  const size_t need = (size_t)tiles * sizeof(int) * (2 + (size_t)cap) + 64;
  CPM_REQUIRE(workspace && workspace_bytes >= need, "workspace too small (cpm_roi_align_fpn_gather_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  int* active_count = (int*)workspace;                       // [16] (one used), then counts, active ids, lists
  int* tile_count = active_count + 16;
  int* active = tile_count + tiles;
  int* tile_list = active + tiles;
  if (hipMemsetAsync(active_count, 0, (size_t)(tiles + 16) * sizeof(int), s) != hipSuccess) return CPM_ELAUNCH;
  for (int i = 0; i < num_levels; ++i)                       // maps that are not accumulated into start from zero
    if (!((accumulate_mask >> i) & 1) &&
        hipMemsetAsync(grad_feats[i], 0, (size_t)B * hs[i] * ws[i] * C * sizeof(float), s) != hipSuccess)
      return CPM_ELAUNCH;
  if (K == 0) return CPM_OK;
  hipLaunchKernelGGL(roi_bin_tiles, dim3(K), dim3(64), 0, s, L, G, S, K, B, cap, tile_count, tile_list);
  {
    int rc = cpm::check_launch("roi_align gather: binning");
    if (rc != CPM_OK) return rc;
  }
  hipLaunchKernelGGL(roi_active_tiles, dim3(cpm::cdiv(tiles, 256)), dim3(256), 0, s, tile_count, tiles, active_count,
                     active);
  hipLaunchKernelGGL(roi_align_bwd_gather, dim3(tiles < 2048 ? tiles : 2048), dim3(GTHREADS), 0, s, L, G, S, C, cap,
                     tile_count, tile_list, active_count, active, accumulate_mask);
  return cpm::check_launch("roi_align gather: tiles");
}

CPM_EXPORT int cpm_roi_align_fpn_backward_gather(const float* grad_output, float* const* grad_feats, const int* hs,
                                                 const int* ws, const float* scales, int num_levels,
                                                 const float* rois, int K, int B, int C, int pooled_h, int pooled_w,
                                                 int sampling_ratio, float k_min, float k_max, float canonical_scale,
                                                 float canonical_level, float eps, int accumulate_mask,
                                                 void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(pooled_h > 0 && pooled_w > 0, "bad shape");
  CPM_REQUIRE(pooled_h <= GBINS && pooled_w <= GBINS, "pooled size above 16: use cpm_roi_align_fpn_backward");
  CPM_REQUIRE(K == 0 || (grad_output && rois), "null pointer");
  RoiSets S = {};
  S.n = 1;
  S.rois[0] = rois; S.grad[0] = grad_output; S.first[0] = 0; S.first[1] = K;
  S.ph[0] = pooled_h; S.pw[0] = pooled_w; S.ratio[0] = sampling_ratio;
  return backward_gather_sets(S, K, grad_feats, hs, ws, scales, num_levels, B, C, k_min, k_max, canonical_scale,
                              canonical_level, eps, accumulate_mask, workspace, workspace_bytes, stream);
}

CPM_EXPORT int cpm_roi_align_fpn_backward_gather_sets(int n_sets, const float* const* grad_outputs,
                                                      const float* const* rois, const int* Ks, const int* pooled_hs,
                                                      const int* pooled_ws, const int* sampling_ratios,
                                                      float* const* grad_feats, const int* hs, const int* ws,
                                                      const float* scales, int num_levels, int B, int C, float k_min,
                                                      float k_max, float canonical_scale, float canonical_level,
                                                      float eps, int accumulate_mask, void* workspace,
                                                      size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(n_sets >= 1 && n_sets <= GSETS, "1 .. 8 RoI sets");
  CPM_REQUIRE(grad_outputs && rois && Ks && pooled_hs && pooled_ws && sampling_ratios, "null pointer");
  RoiSets S = {};
  int total = 0;
  for (int i = 0; i < n_sets; ++i) {
    CPM_REQUIRE(Ks[i] >= 0 && pooled_hs[i] > 0 && pooled_ws[i] > 0 && pooled_hs[i] <= GBINS && pooled_ws[i] <= GBINS,
                "bad set (pooled size 1 .. 16)");
    if (Ks[i] == 0) continue;                                 // an empty set takes no slot
    CPM_REQUIRE(grad_outputs[i] && rois[i], "null set pointer");
    CPM_REQUIRE((((uintptr_t)grad_outputs[i]) & 15) == 0, "gradients must be 16-byte aligned");
    const int k = S.n++;
    S.rois[k] = rois[i]; S.grad[k] = grad_outputs[i];
    S.first[k] = total;
    S.ph[k] = pooled_hs[i]; S.pw[k] = pooled_ws[i]; S.ratio[k] = sampling_ratios[i];
    total += Ks[i];
  }
  S.first[S.n] = total;
  if (S.n == 0) S.n = 1;                                      // all empty: K == 0 below clears the fresh maps only
  return backward_gather_sets(S, total, grad_feats, hs, ws, scales, num_levels, B, C, k_min, k_max, canonical_scale,
                              canonical_level, eps, accumulate_mask, workspace, workspace_bytes, stream);
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
