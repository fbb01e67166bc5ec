// HBM-bound helper kernels of the conv/GN stacks: epilogue backward, im2col for the stem,
// max-pool, GroupNorm(+ReLU) forward/backward, FPN top-down backward, fused SGD.  NHWC, fp32, gfx950.
#include "common.h"
#include "igemm_common.h"

namespace {

int grid_for(int64_t total, int per_block = 256, int cap = 4096) {
  int64_t b = (total + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- backward of the fused conv epilogue ---------------------------------------------------------------
// y = relu?(conv*scale + shift [+ residual]).  dpre = dy * [y>0] * scale ; dshift[k] += sum_m dy*[y>0].
// grid (ceil(K/64), slabs); block 256 = 4 waves; lane = channel, waves split the slab's rows.
__global__ __launch_bounds__(256) void epilogue_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ scale, int relu, int64_t M,
                                                           int K, int rows_per_slab, float* __restrict__ dpre,
                                                           float* __restrict__ dres, float* __restrict__ dshift) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < M ? r0 + rows_per_slab : M;
  float acc = 0.f;
  if (k < K) {
    const float sc = scale ? scale[k] : 1.f;
    for (int64_t m = r0 + wave; m < r1; m += 4) {
      const int64_t idx = m * K + k;
      float g = dy[idx];
      if (relu && !(y[idx] > 0.f)) g = 0.f;
      acc += g;
      if (dres) dres[idx] = g;
      if (dpre) dpre[idx] = g * sc;
    }
  }
  if (dshift) {
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && k < K) atomicAdd(dshift + k, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
  }
}

// Vectorised variant: a thread owns 4 consecutive channels (one float4) and walks rows; the 256 threads of a block
// cover `RP` rows x `CV` float4-columns per iteration (CV = 256/RP), so every load/store is 16 bytes and coalesced.
template <int RP>
__global__ __launch_bounds__(256) void epilogue_bwd_vec_kernel(const float* __restrict__ dy,
                                                               const float* __restrict__ y,
                                                               const float* __restrict__ scale, int relu, int64_t M,
                                                               int K, int rows_per_slab, float* __restrict__ dpre,
                                                               float* __restrict__ dres, float* __restrict__ dshift) {
  constexpr int CV = 256 / RP;                    // float4 columns handled by one block
  __shared__ float4 part[256];
  const int cv = threadIdx.x % CV, rp = threadIdx.x / CV;
  const int k = (blockIdx.x * CV + cv) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < M ? r0 + rows_per_slab : M;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (scale) sc = *(const float4*)(scale + k);
    int64_t m = r0 + rp;
    // four independent rows per trip: their loads are all in flight before the first use
    for (; m + 3 * RP < r1; m += 4 * RP) {
      float4 g[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = *(const float4*)(dy + (m + u * RP) * K + k);
      if (relu) {
#pragma unroll
        for (int u = 0; u < 4; ++u) yv[u] = *(const float4*)(y + (m + u * RP) * K + k);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t idx = (m + u * RP) * K + k;
        if (relu) {
          if (!(yv[u].x > 0.f)) g[u].x = 0.f;
          if (!(yv[u].y > 0.f)) g[u].y = 0.f;
          if (!(yv[u].z > 0.f)) g[u].z = 0.f;
          if (!(yv[u].w > 0.f)) g[u].w = 0.f;
        }
        acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w;
        if (dres) *(float4*)(dres + idx) = g[u];
        if (dpre) *(float4*)(dpre + idx) = make_float4(g[u].x * sc.x, g[u].y * sc.y, g[u].z * sc.z, g[u].w * sc.w);
      }
    }
    for (; m < r1; m += RP) {
      const int64_t idx = m * K + k;
      float4 g = *(const float4*)(dy + idx);
      if (relu) {
        const float4 yv = *(const float4*)(y + idx);
        if (!(yv.x > 0.f)) g.x = 0.f;
        if (!(yv.y > 0.f)) g.y = 0.f;
        if (!(yv.z > 0.f)) g.z = 0.f;
        if (!(yv.w > 0.f)) g.w = 0.f;
      }
      acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      if (dres) *(float4*)(dres + idx) = g;
      if (dpre) *(float4*)(dpre + idx) = make_float4(g.x * sc.x, g.y * sc.y, g.z * sc.z, g.w * sc.w);
    }
  }
  if (dshift) {
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rp == 0 && k < K) {
      float4 t = acc;
      for (int r = 1; r < RP; ++r) {
        const float4 o = part[r * CV + cv];
        t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
      }
      atomicAdd(dshift + k, t.x); atomicAdd(dshift + k + 1, t.y);
      atomicAdd(dshift + k + 2, t.z); atomicAdd(dshift + k + 3, t.w);
    }
  }
}

// ---- im2col (thin-channel stem) --------------------------------------------------------------------------
__global__ void im2col_kernel(const float* __restrict__ x, int layout, int N, int C, int H, int W, int R, int S,
                              int stride, int pad, int P, int Q, int Kpad, float* __restrict__ out) {
  const int64_t total = (int64_t)N * P * Q * Kpad;
  const int K = R * S * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int col = idx % Kpad;
    const int64_t m = idx / Kpad;
    float v = 0.f;
    if (col < K) {
      const int c = col % C, t = col / C;
      const int s = t % S, r = t / S;
      const int q = m % Q;
      const int64_t t2 = m / Q;
      const int p = t2 % P, n = t2 / P;
      const int ih = p * stride - pad + r, iw = q * stride - pad + s;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
        v = layout == CPM_LAYOUT_NHWC ? x[(((int64_t)n * H + ih) * W + iw) * C + c]
                                      : x[(((int64_t)n * C + c) * H + ih) * W + iw];
    }
    out[idx] = v;
  }
}

// The same with Kpad % 4 == 0 and Kpad <= 512 (the 7x7x3 stem: 147 -> 160 columns): the (tap row, tap column,
// channel) of every column comes from a table built once per workgroup instead of five divisions per element, a
// thread produces four consecutive columns of one row and stores them as one float4 (the kernel above spent its
// time in integer division: 281 us for the 344 MB of the 2 x 800 x 1344 stem, HBM floor ~70 us).
__global__ void __launch_bounds__(256) im2col_vec_kernel(const float* __restrict__ x, int layout, int N, int C, int H,
                                                         int W, int R, int S, int stride, int pad, int P, int Q,
                                                         int Kpad, float* __restrict__ out) {
  __shared__ int s_r[512], s_s[512], s_c[512];
  const int K = R * S * C;
  for (int col = threadIdx.x; col < Kpad; col += blockDim.x) {
    const int c = col % C, t = col / C;
    s_c[col] = col < K ? c : -1;
    s_s[col] = t % S;
    s_r[col] = t / S;
  }
  __syncthreads();
  const int groups = Kpad / 4;
  const int64_t rows = (int64_t)N * P * Q, total = rows * groups;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int gidx = (int)(idx % groups);
    const int64_t m = idx / groups;
    const int q = (int)(m % Q);
    const int64_t t2 = m / Q;
    const int p = (int)(t2 % P), n = (int)(t2 / P);
    const int ih0 = p * stride - pad, iw0 = q * stride - pad;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = gidx * 4 + k;
      const int c = s_c[col];
      const int ih = ih0 + s_r[col], iw = iw0 + s_s[col];
      v[k] = 0.f;
      if (c >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
        v[k] = layout == CPM_LAYOUT_NHWC ? x[(((int64_t)n * H + ih) * W + iw) * C + c]
                                         : x[(((int64_t)n * C + c) * H + ih) * W + iw];
    }
    *(float4*)(out + m * Kpad + gidx * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- 3x3 / stride 2 / pad 1 max-pool, NHWC (4 channels per thread) ---------------------------------------
__global__ void maxpool_kernel(const float* __restrict__ x, int N, int H, int W, int C, int P, int Q,
                               float* __restrict__ y) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * P * Q * C4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (idx % C4) * 4;
    int64_t t = idx / C4;
    const int q = t % Q; t /= Q;
    const int p = t % P;
    const int n = t / P;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int r = 0; r < 3; ++r) {
      const int ih = p * 2 - 1 + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int s = 0; s < 3; ++s) {
        const int iw = q * 2 - 1 + s;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4 v = *(const float4*)(x + (((int64_t)n * H + ih) * W + iw) * C + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *(float4*)(y + (((int64_t)n * P + p) * Q + q) * C + c) = m;
  }
}

// ---- GroupNorm (+ReLU) ------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
  // 256 threads
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// grid (G, N), block 256: one workgroup per (sample, group); two-pass moments like a careful CPU loop
__global__ __launch_bounds__(256) void gn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int HW, int C, int G, float eps,
                                                     int relu, float* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out) {
  __shared__ float red[4];
  const int g = blockIdx.x, n = blockIdx.y;
  const int Cg = C / G;
  const int cnt = HW * Cg;
  const float* xb = x + (size_t)n * HW * C + g * Cg;
  float s = 0.f;
  for (int i = threadIdx.x; i < cnt; i += 256) s += xb[(size_t)(i / Cg) * C + (i % Cg)];
  const float mean = block_sum(s, red) / (float)cnt;
  float v = 0.f;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const float d = xb[(size_t)(i / Cg) * C + (i % Cg)] - mean;
    v += d * d;
  }
  const float var = block_sum(v, red) / (float)cnt;
  const float rstd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[n * G + g] = mean;
    rstd_out[n * G + g] = rstd;
  }
  float* yb = y + (size_t)n * HW * C + g * Cg;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const int c = i % Cg;
    const size_t off = (size_t)(i / Cg) * C + c;
    float o = (xb[off] - mean) * rstd * gamma[g * Cg + c] + beta[g * Cg + c];
    if (relu) o = fmaxf(o, 0.f);
    yb[off] = o;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ y, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, int HW, int C, int G,
                                                     int relu, float* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta) {
  __shared__ float red[4];
  __shared__ float sg[256], sb[256];      // per-channel partials (Cg <= 256)
  const int g = blockIdx.x, n = blockIdx.y;
  const int Cg = C / G;
  const int cnt = HW * Cg;
  const size_t base = (size_t)n * HW * C + g * Cg;
  const float mean = mean_in[n * G + g], rstd = rstd_in[n * G + g];
  for (int i = threadIdx.x; i < Cg; i += 256) { sg[i] = 0.f; sb[i] = 0.f; }
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  // When Cg divides 256 a thread meets the same channel on every trip (i % Cg == threadIdx.x % Cg): the per-channel
  // sums stay in registers and are folded through LDS once; LDS float atomics (ds_add_f32 costs ~64 cycles per wave
  // instruction, more with 256 threads on Cg addresses) are only the fallback for odd group widths.
  const bool fixed_c = (256 % Cg) == 0;
  float pg = 0.f, pb = 0.f;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const int c = i % Cg;
    const size_t off = base + (size_t)(i / Cg) * C + c;
    float go = dy[off];
    if (relu && !(y[off] > 0.f)) go = 0.f;
    const float xh = (x[off] - mean) * rstd;
    const float dxh = go * gamma[g * Cg + c];
    s1 += dxh;
    s2 += dxh * xh;
    if (fixed_c) {
      pg += go * xh;
      pb += go;
    } else {
      atomicAdd(&sg[c], go * xh);
      atomicAdd(&sb[c], go);
    }
  }
  if (fixed_c) {
    __shared__ float tg[256], tb[256];
    tg[threadIdx.x] = pg;
    tb[threadIdx.x] = pb;
    __syncthreads();
    if (threadIdx.x < Cg) {
      float a = 0.f, b = 0.f;
      for (int j = threadIdx.x; j < 256; j += Cg) { a += tg[j]; b += tb[j]; }
      sg[threadIdx.x] = a;
      sb[threadIdx.x] = b;
    }
  }
  const float m1 = block_sum(s1, red) / (float)cnt;
  const float m2 = block_sum(s2, red) / (float)cnt;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const int c = i % Cg;
    const size_t off = base + (size_t)(i / Cg) * C + c;
    float go = dy[off];
    if (relu && !(y[off] > 0.f)) go = 0.f;
    const float xh = (x[off] - mean) * rstd;
    dx[off] = rstd * (go * gamma[g * Cg + c] - m1 - xh * m2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cg; i += 256) {
    if (dgamma) atomicAdd(dgamma + g * Cg + i, sg[i]);
    if (dbeta) atomicAdd(dbeta + g * Cg + i, sb[i]);
  }
}

// GroupNorm backward for SMALL groups (the grid head: GroupNorm(36, 576) on 7x7 maps = 49 pixels x 16 channels): one
// WAVEFRONT per (sample, group) instead of a 256-thread workgroup -- a lane owns up to TRIPS float4 (pixel, channel
// quad) slots, keeps the gated gradient and the normalised input of all of them in registers between the moment
// pass and the output pass (the workgroup version re-read dy, y and x), and the two moments are wave reductions (DPP
// row prefix + four v_readlane) with no barrier.  A lane meets the same channel quad on every trip (the quads per
// pixel divide 64), so the gamma / beta partials stay in registers and are folded once through a private LDS slab.
__device__ __forceinline__ float wave_total(float v) {
  int x = __float_as_int(v);
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true)));
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true)));
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true)));
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true)));
  return (__int_as_float(__builtin_amdgcn_readlane(x, 15)) + __int_as_float(__builtin_amdgcn_readlane(x, 31))) +
         (__int_as_float(__builtin_amdgcn_readlane(x, 47)) + __int_as_float(__builtin_amdgcn_readlane(x, 63)));
}

// THREADS = 64: one wavefront per (sample, group), four of them per workgroup; THREADS = 256: the whole workgroup on one
// (sample, group) for the wider groups (GroupNorm(9, 576) on the 14x14 deconvolution output: 196 pixels x 64 channels
// = 13 float4 per thread), same register-resident single pass, moments folded across the four waves through LDS.
template <int THREADS, int TRIPS>
__global__ __launch_bounds__(256) void gn_bwd_wave_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, int N, int HW, int C,
                                                          int G, int qshift, int relu, float* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ zero_out) {
  __shared__ float part[256][9];                        // [thread][4 dgamma + 4 dbeta], padded
  __shared__ float red2[2][4];
  constexpr bool WAVE = THREADS == 64;
  const int wave = threadIdx.x >> 6;
  const int lane = WAVE ? (threadIdx.x & 63) : threadIdx.x;      // index inside the job's thread set
  const int job = WAVE ? blockIdx.x * 4 + wave : blockIdx.x;
  if (job >= N * G) return;                             // WAVE: no block-wide barrier below; else block-uniform
  const int n = job / G, g = job - n * G;
  const int Cg = C / G, Q = 1 << qshift, slots = HW << qshift;
  const int quad = lane & (Q - 1);
  const size_t base = (size_t)n * HW * C + (size_t)g * Cg + 4 * quad;
  const float mean = mean_in[job], rstd = rstd_in[job];
  const float4 gm = *(const float4*)(gamma + g * Cg + 4 * quad);
  float4 go[TRIPS], xh[TRIPS];
  float s1 = 0.f, s2 = 0.f;
  float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pb = pg;
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int slot = lane + THREADS * t;
    go[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    xh[t] = go[t];
    if (slot < slots) {
      const size_t off = base + (size_t)(slot >> qshift) * C;
      float4 d = *(const float4*)(dy + off);
      const float4 xv = *(const float4*)(x + off);
      if (relu) {
        const float4 yv = *(const float4*)(y + off);
        d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
        d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
      }
      const float4 h = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd,
                                   (xv.w - mean) * rstd);
      go[t] = d; xh[t] = h;
      const float a0 = d.x * gm.x, a1 = d.y * gm.y, a2 = d.z * gm.z, a3 = d.w * gm.w;
      s1 += (a0 + a1) + (a2 + a3);
      s2 += (a0 * h.x + a1 * h.y) + (a2 * h.z + a3 * h.w);
      pg.x += d.x * h.x; pg.y += d.y * h.y; pg.z += d.z * h.z; pg.w += d.w * h.w;
      pb.x += d.x; pb.y += d.y; pb.z += d.z; pb.w += d.w;
    }
  }
  const float inv = 1.f / (float)(HW * Cg);
  float m1 = wave_total(s1), m2 = wave_total(s2);
  if (!WAVE) {
    if ((threadIdx.x & 63) == 0) { red2[0][wave] = m1; red2[1][wave] = m2; }
    __syncthreads();
    m1 = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
    m2 = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
  }
  m1 *= inv; m2 *= inv;
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int slot = lane + THREADS * t;
    if (slot < slots) {
      const float4 d = go[t], h = xh[t];
      float4 o;
      o.x = rstd * (d.x * gm.x - m1 - h.x * m2); o.y = rstd * (d.y * gm.y - m1 - h.y * m2);
      o.z = rstd * (d.z * gm.z - m1 - h.z * m2); o.w = rstd * (d.w * gm.w - m1 - h.w * m2);
      *(float4*)(dx + base + (size_t)(slot >> qshift) * C) = o;
      // side job for a layer chain (cpm_groupnorm_backward_zero): the conv's INPUT gradient -- same shape -- cleared
      // for the reduction-split data-gradient launch that follows
      if (zero_out) *(float4*)(zero_out + base + (size_t)(slot >> qshift) * C) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (!dgamma && !dbeta) return;
  float* pr = &part[threadIdx.x][0];
  pr[0] = pg.x; pr[1] = pg.y; pr[2] = pg.z; pr[3] = pg.w;
  pr[4] = pb.x; pr[5] = pb.y; pr[6] = pb.z; pr[7] = pb.w;
  if (WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
  if (lane < Cg) {                                      // channel `lane` of the group: quad lane >> 2, component lane & 3
    const int q = lane >> 2, k = lane & 3;
    const int first = WAVE ? wave * 64 : 0;
    float a = 0.f, b = 0.f;
    for (int l = q; l < THREADS; l += Q) { a += part[first + l][k]; b += part[first + l][4 + k]; }
    if (dgamma) atomicAdd(dgamma + g * Cg + lane, a);
    if (dbeta) atomicAdd(dbeta + g * Cg + lane, b);
  }
}

// Forward with the same ownership (see gn_bwd_wave_kernel): x is read ONCE; mean and the centred second moment are
// taken from the register copy (the two-pass formula of gn_fwd_kernel, without its two extra reads).
template <int THREADS, int TRIPS>
__global__ __launch_bounds__(256) void gn_fwd_wave_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int N, int HW, int C, int G,
                                                          int qshift, float eps, int relu, float* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          float* __restrict__ fill_out,
                                                          const float* __restrict__ fill_bias) {
  __shared__ float red2[2][4];
  constexpr bool WAVE = THREADS == 64;
  const int wave = threadIdx.x >> 6;
  const int lane = WAVE ? (threadIdx.x & 63) : threadIdx.x;
  const int job = WAVE ? blockIdx.x * 4 + wave : blockIdx.x;
  if (job >= N * G) return;                             // WAVE: no block-wide barrier below; else block-uniform
  const int n = job / G, g = job - n * G;
  const int Cg = C / G, Q = 1 << qshift, slots = HW << qshift;
  const int quad = lane & (Q - 1);
  const size_t base = (size_t)n * HW * C + (size_t)g * Cg + 4 * quad;
  float4 xv[TRIPS];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int slot = lane + THREADS * t;
    xv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot < slots) {
      xv[t] = *(const float4*)(x + base + (size_t)(slot >> qshift) * C);
      s += (xv[t].x + xv[t].y) + (xv[t].z + xv[t].w);
    }
  }
  const float inv = 1.f / (float)(HW * Cg);
  float tot = wave_total(s);
  if (!WAVE) {
    if ((threadIdx.x & 63) == 0) red2[0][wave] = tot;
    __syncthreads();
    tot = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
  }
  const float mean = tot * inv;
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int slot = lane + THREADS * t;
    if (slot < slots) {
      const float a = xv[t].x - mean, b = xv[t].y - mean, c = xv[t].z - mean, d = xv[t].w - mean;
      v += (a * a + b * b) + (c * c + d * d);
    }
  }
  float vt = wave_total(v);
  if (!WAVE) {
    if ((threadIdx.x & 63) == 0) red2[1][wave] = vt;
    __syncthreads();
    vt = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
  }
  const float rstd = 1.f / sqrtf(vt * inv + eps);
  if (lane == 0) { mean_out[job] = mean; rstd_out[job] = rstd; }
  const float4 gm = *(const float4*)(gamma + g * Cg + 4 * quad);
  const float4 bt = *(const float4*)(beta + g * Cg + 4 * quad);
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int slot = lane + THREADS * t;
    if (slot < slots) {
      float4 o;
      o.x = (xv[t].x - mean) * rstd * gm.x + bt.x; o.y = (xv[t].y - mean) * rstd * gm.y + bt.y;
      o.z = (xv[t].z - mean) * rstd * gm.z + bt.z; o.w = (xv[t].w - mean) * rstd * gm.w + bt.w;
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *(float4*)(y + base + (size_t)(slot >> qshift) * C) = o;
    }
  }
  // side job for a layer chain (cpm_groupnorm_forward_fill): the NEXT convolution's output tensor -- same [N, HW, C]
  // shape -- leaves here holding its bias (or zeros), so that a reduction-split launch adds into it without a seed /
  // clear launch of its own
  if (fill_out) {
    const float4 fb = fill_bias ? *(const float4*)(fill_bias + g * Cg + 4 * quad) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int slot = lane + THREADS * t;
      if (slot < slots) *(float4*)(fill_out + base + (size_t)(slot >> qshift) * C) = fb;
    }
  }
}

// ---- FPN top-down backward -----------------------------------------------------------------------------------
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, int N, int P, int Q, int C, int accumulate,
                                    float* __restrict__ dtop) {
  const int TP = (P + 1) / 2, TQ = (Q + 1) / 2, C4 = C >> 2;
  const int64_t total = (int64_t)N * TP * TQ * C4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (idx % C4) * 4;
    int64_t t = idx / C4;
    const int q = t % TQ; t /= TQ;
    const int p = t % TP;
    const int n = t / TP;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const int hh = 2 * p + a, ww = 2 * q + b;
        if (hh < P && ww < Q) {
          const float4 v = *(const float4*)(dy + (((int64_t)n * P + hh) * Q + ww) * C + c);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
      }
    float4* d = (float4*)(dtop + (((int64_t)n * TP + p) * TQ + q) * C + c);
    if (accumulate) { float4 o = *d; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *d = s;
  }
}

// ---- fused SGD ------------------------------------------------------------------------------------------------
// One thread updates 4 consecutive elements; the segment of each 64-element block comes from a lookup table
// (segments are 64-element aligned; blocks in the alignment gaps are skipped); the segment's parameter group gives
// the learning rate / weight decay, which travel as kernel arguments (no host->device copy when the schedule moves).
struct SgdGroups { float lr[8]; float wd[8]; };

// w4 (or null): the pre-split image of the updated parameters (cpm_split_w4's format), written with them
// zero_g: the gradient is cleared behind its use (the next step's zero_grad for free: the kernel has the line anyway)
__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                           const int32_t* __restrict__ block_seg, const int64_t* __restrict__ seg_end,
                           const int32_t* __restrict__ seg_group, SgdGroups grp, int64_t total, float momentum,
                           float grad_scale, int first_step, uint4* __restrict__ w4, int64_t range_begin, int zero_g) {
  const int64_t nvec = total >> 2;            // (total = end of the range to update, range_begin its first element)
  for (int64_t v = (range_begin >> 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
       v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = v << 2;
    const int sidx = block_seg[i >> 6];
    if (sidx < 0) continue;
    const int64_t end = seg_end[sidx];
    if (i >= end) continue;
    const int gi = seg_group[sidx] & 7;
    const float lr = grp.lr[gi], wd = grp.wd[gi];
    float4 pv = *(const float4*)(p + i);
    const float4 gv = *(const float4*)(g + i);
    float4 bv = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(buf + i);
    float* pp = &pv.x; const float* gp = &gv.x; float* bp = &bv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = gp[k] * grad_scale;
      if (wd != 0.f) d = d + wd * pp[k];
      const float b = first_step ? d : momentum * bp[k] + d;
      bp[k] = b;
      pp[k] = pp[k] - lr * b;
    }
    if (i + 4 <= end) {
      *(float4*)(p + i) = pv;
      *(float4*)(buf + i) = bv;
      if (zero_g) *(float4*)(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int k = 0; k < 4 && i + k < end; ++k) {
        p[i + k] = pp[k]; buf[i + k] = bp[k];
        if (zero_g) g[i + k] = 0.f;
      }
    }
    if (w4) {                                   // (a tensor's last partial quad: its image is never read as a quad)
      uint2 hi, lo;
      cpmconv::split4(pv, hi, lo);
      w4[v] = make_uint4(hi.x, hi.y, lo.x, lo.y);
    }
  }
}

// ---- data gradient of the RPN's two 1x1 predictors, all levels, one launch -----------------------------------------
// rpn/rpn.py:24-31: cls_logits (A channels) and bbox_pred (4A channels) read the same ReLU'd map t.  Their data gradients
// have reductions of 3 and 12: as implicit GEMMs they are two passes over the 137 MB gradient of P2's map at 2-6 TFLOP/s
// (223 us; the second one adds into the first one's output).  Here: dx = [t > 0] * (dy_cls W_cls + dy_box W_box) for
// every pixel of every level, one float4 of channels per thread, the 5A x C weights in LDS -- one read of t, one write
// of dx.
struct PredLevels {
  const float* dy_cls[8];
  const float* dy_box[8];
  const float* t[8];
  float* dx[8];
  int64_t pix_end[8];          // cumulative pixel counts
  int n;
};

__global__ __launch_bounds__(256) void rpn_pred_dgrad_kernel(PredLevels L, const float* __restrict__ w_cls,
                                                             const float* __restrict__ w_box, int A, int C, int gate) {
  extern __shared__ float pw[];                        // [5A][C]
  for (int i = threadIdx.x; i < 5 * A * C; i += 256) pw[i] = i < A * C ? w_cls[i] : w_box[i - A * C];
  __syncthreads();
  const int q = C >> 2, ppb = 256 / q;
  const int cq = threadIdx.x % q, pl = threadIdx.x / q;
  const int64_t total = L.pix_end[L.n - 1];
  for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < total; p += (int64_t)gridDim.x * ppb) {
    int lv = 0;
    while (p >= L.pix_end[lv]) ++lv;
    const int64_t pi = p - (lv ? L.pix_end[lv - 1] : 0);
    const float* dc = L.dy_cls[lv] + pi * A;
    const float* db = L.dy_box[lv] + pi * 4 * A;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < A; ++k) {
      const float g = dc[k];
      const float4 w = *(const float4*)&pw[k * C + 4 * cq];
      acc.x += g * w.x; acc.y += g * w.y; acc.z += g * w.z; acc.w += g * w.w;
    }
    for (int k = 0; k < 4 * A; ++k) {
      const float g = db[k];
      const float4 w = *(const float4*)&pw[(A + k) * C + 4 * cq];
      acc.x += g * w.x; acc.y += g * w.y; acc.z += g * w.z; acc.w += g * w.w;
    }
    if (gate) {
      const float4 tv = *(const float4*)(L.t[lv] + pi * C + 4 * cq);
      acc.x = tv.x > 0.f ? acc.x : 0.f; acc.y = tv.y > 0.f ? acc.y : 0.f;
      acc.z = tv.z > 0.f ? acc.z : 0.f; acc.w = tv.w > 0.f ? acc.w : 0.f;
    }
    *(float4*)(L.dx[lv] + pi * C + 4 * cq) = acc;
  }
}

}  // namespace

CPM_EXPORT int cpm_epilogue_backward(const float* dy, const float* y, const float* scale, int relu, int64_t M,
                                     int K, float* dpre, float* dres, float* dshift, void* stream) {
  CPM_REQUIRE(M >= 0 && K > 0, "bad shape");
  if (M == 0) return CPM_OK;
  CPM_REQUIRE(dy && (!relu || y), "null pointer");
  const bool aligned = (K % 4 == 0) && !((uintptr_t)dy & 15) && !((uintptr_t)y & 15) && !((uintptr_t)dpre & 15) &&
                       !((uintptr_t)dres & 15) && !((uintptr_t)scale & 15);
  const int cvk = K / 4;                      // float4 per row
  if (aligned && cvk >= 16) {
    // rows per block iteration: as many as fit 256 threads (power of two), at least 1
    int rp = 1;
    while (rp < 16 && cvk * rp * 2 <= 256) rp *= 2;
    // The bias gradient ends in one float atomic per (block, channel): with ~2000 row slabs that is ~2000 adds queued
    // on each of a few hundred addresses (measured 220 us for a 138 MB map that streams in 35 us).  When dshift is
    // wanted, blocks are made narrow (16 float4 columns) so that the same number of blocks needs 4-16x fewer slabs.
    if (dshift) rp = 16;
    const int cv = 256 / rp;
    const int gx = cpm::cdiv(cvk, cv);
    int64_t want = (int64_t)(dshift ? 1024 : 2048) / gx;        // ~4-8 blocks per CU overall
    if (want < 1) want = 1;
    int64_t rows = (M + want - 1) / want;
    if (rows < 64) rows = 64;
    const int slabs = (int)((M + rows - 1) / rows);
#define EL(RP)                                                                                               \
  hipLaunchKernelGGL((epilogue_bwd_vec_kernel<RP>), dim3(gx, slabs), dim3(256), 0, (hipStream_t)stream, dy, y, \
                     scale, relu, M, K, (int)rows, dpre, dres, dshift)
    if (rp == 1) EL(1); else if (rp == 2) EL(2); else if (rp == 4) EL(4); else if (rp == 8) EL(8); else EL(16);
#undef EL
    return cpm::check_launch("epilogue_backward");
  }
  int slabs = (int)((M + 511) / 512);
  if (slabs > 2048) slabs = 2048;
  const int rows = (int)((M + slabs - 1) / slabs);
  slabs = (int)((M + rows - 1) / rows);
  hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(cpm::cdiv(K, 64), slabs), dim3(256), 0, (hipStream_t)stream, dy, y,
                     scale, relu, M, K, rows, dpre, dres, dshift);
  return cpm::check_launch("epilogue_backward");
}

CPM_EXPORT int cpm_im2col(const float* x, int layout, int N, int C, int H, int W, int R, int S, int stride, int pad,
                          int P, int Q, int Kpad, float* out, void* stream) {
  CPM_REQUIRE(x && out, "null pointer");
  CPM_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && R > 0 && S > 0 && stride > 0 && Kpad >= R * S * C, "bad shape");
  CPM_REQUIRE(P == (H + 2 * pad - R) / stride + 1 && Q == (W + 2 * pad - S) / stride + 1, "bad output size");
  const int64_t total = (int64_t)N * P * Q * Kpad;
  if (Kpad % 4 == 0 && Kpad <= 512 && (((uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(im2col_vec_kernel, dim3(grid_for(total / 4, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x,
                       layout, N, C, H, W, R, S, stride, pad, P, Q, Kpad, out);
  else
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, layout,
                       N, C, H, W, R, S, stride, pad, P, Q, Kpad, out);
  return cpm::check_launch("im2col");
}

CPM_EXPORT int cpm_maxpool3x3s2_forward(const float* x, int N, int H, int W, int C, int P, int Q, float* y,
                                        void* stream) {
  CPM_REQUIRE(x && y, "null pointer");
  CPM_REQUIRE(C % 4 == 0, "C must be a multiple of 4");
  CPM_REQUIRE(P == (H + 2 - 3) / 2 + 1 && Q == (W + 2 - 3) / 2 + 1, "bad output size");
  const int64_t total = (int64_t)N * P * Q * (C / 4);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, N, H, W,
                     C, P, Q, y);
  return cpm::check_launch("maxpool");
}

__global__ __launch_bounds__(256) void fill_rows4_kernel(float4* __restrict__ out, const float4* __restrict__ bias4,
                                                         int64_t total4, int c4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256)
    out[i] = bias4 ? bias4[i % c4] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void fill_rows1_kernel(float* __restrict__ out, const float* __restrict__ bias,
                                                         int64_t total, int c) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    out[i] = bias ? bias[i % c] : 0.f;
}

static int groupnorm_forward_impl(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int G,
                                  float eps, int relu, float* y, float* mean, float* rstd, float* fill_out,
                                  const float* fill_bias, void* stream) {
  CPM_REQUIRE(N >= 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "bad shape");
  if (N == 0) return CPM_OK;
  CPM_REQUIRE(x && gamma && beta && y && mean && rstd, "null pointer");
  const int Cg = C / G, Q = Cg / 4;
  const bool pow2 = Q > 0 && (Q & (Q - 1)) == 0;
  const bool vec = (Cg & 3) == 0 && (C & 3) == 0 && pow2 && Q <= 64 && (int64_t)N * G < (1ll << 30) &&
                   ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)fill_out |
                      (uintptr_t)fill_bias) & 15) == 0);
  int qshift = 0;
  while ((1 << qshift) < Q) ++qshift;
  if (vec && (int64_t)HW * Q <= 64 * 4) {
    hipLaunchKernelGGL((gn_fwd_wave_kernel<64, 4>), dim3((unsigned)(((int64_t)N * G + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, gamma, beta, N, HW, C, G, qshift, eps, relu, y, mean, rstd, fill_out,
                       fill_bias);
    return cpm::check_launch("groupnorm_forward (wave per group)");
  }
  if (vec && (int64_t)HW * Q <= 256 * 13) {
    hipLaunchKernelGGL((gn_fwd_wave_kernel<256, 13>), dim3((unsigned)((int64_t)N * G)), dim3(256), 0,
                       (hipStream_t)stream, x, gamma, beta, N, HW, C, G, qshift, eps, relu, y, mean, rstd, fill_out,
                       fill_bias);
    return cpm::check_launch("groupnorm_forward (workgroup per group, single pass)");
  }
  hipLaunchKernelGGL(gn_fwd_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, HW, C, G, eps, relu,
                     y, mean, rstd);
  if (fill_out) {                                            // (the general kernel has no side job: a launch of its own)
    if ((C & 3) == 0 && ((((uintptr_t)fill_out | (uintptr_t)fill_bias) & 15) == 0)) {
      const int64_t total4 = (int64_t)N * HW * C / 4;
      hipLaunchKernelGGL(fill_rows4_kernel, dim3((unsigned)(total4 / 256 + 1 > 4096 ? 4096 : total4 / 256 + 1)), dim3(256),
                         0, (hipStream_t)stream, (float4*)fill_out, (const float4*)fill_bias, total4, C / 4);
    } else {                                                 // any channel count / alignment: one element per thread
      const int64_t total = (int64_t)N * HW * C;
      hipLaunchKernelGGL(fill_rows1_kernel, dim3((unsigned)(total / 256 + 1 > 4096 ? 4096 : total / 256 + 1)), dim3(256), 0,
                         (hipStream_t)stream, fill_out, fill_bias, total, C);
    }
  }
  return cpm::check_launch("groupnorm_forward");
}

CPM_EXPORT int cpm_groupnorm_forward(const float* x, const float* gamma, const float* beta, int N, int HW, int C,
                                     int G, float eps, int relu, float* y, float* mean, float* rstd, void* stream) {
  return groupnorm_forward_impl(x, gamma, beta, N, HW, C, G, eps, relu, y, mean, rstd, nullptr, nullptr, stream);
}

CPM_EXPORT int cpm_groupnorm_forward_fill(const float* x, const float* gamma, const float* beta, int N, int HW, int C,
                                          int G, float eps, int relu, float* y, float* mean, float* rstd,
                                          float* fill_out, const float* fill_bias, void* stream) {
  CPM_REQUIRE(fill_out, "null fill tensor");
  return groupnorm_forward_impl(x, gamma, beta, N, HW, C, G, eps, relu, y, mean, rstd, fill_out, fill_bias, stream);
}

static int groupnorm_backward_impl(const float* dy, const float* x, const float* y, const float* gamma,
                                   const float* mean, const float* rstd, int N, int HW, int C, int G, int relu,
                                   float* dx, float* dgamma, float* dbeta, float* zero_out, void* stream) {
  CPM_REQUIRE(N >= 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && C / G <= 256, "bad shape");
  if (N == 0) return CPM_OK;
  CPM_REQUIRE(dy && x && gamma && mean && rstd && dx && (!relu || y), "null pointer");
  const int Cg = C / G, Q = Cg / 4;
  const bool pow2 = Q > 0 && (Q & (Q - 1)) == 0;
  const bool al16 = ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)gamma |
                       (uintptr_t)zero_out) & 15) == 0);
  if ((Cg & 3) == 0 && (C & 3) == 0 && pow2 && Q <= 16 && (int64_t)HW * Q <= 64 * 4 && (int64_t)N * G < (1ll << 30) &&
      al16) {
    int qshift = 0;
    while ((1 << qshift) < Q) ++qshift;
    hipLaunchKernelGGL((gn_bwd_wave_kernel<64, 4>), dim3((unsigned)(((int64_t)N * G + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, dy, x, y, gamma, mean, rstd, N, HW, C, G, qshift, relu, dx, dgamma, dbeta,
                       zero_out);
    return cpm::check_launch("groupnorm_backward (wave per group)");
  }
  if ((Cg & 3) == 0 && (C & 3) == 0 && pow2 && Q <= 16 && (int64_t)HW * Q <= 256 * 13 && (int64_t)N * G < (1ll << 30) &&
      al16) {
    int qshift = 0;
    while ((1 << qshift) < Q) ++qshift;
    hipLaunchKernelGGL((gn_bwd_wave_kernel<256, 13>), dim3((unsigned)((int64_t)N * G)), dim3(256), 0,
                       (hipStream_t)stream, dy, x, y, gamma, mean, rstd, N, HW, C, G, qshift, relu, dx, dgamma, dbeta,
                       zero_out);
    return cpm::check_launch("groupnorm_backward (workgroup per group, single pass)");
  }
  hipLaunchKernelGGL(gn_bwd_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream, dy, x, y, gamma, mean, rstd, HW, C,
                     G, relu, dx, dgamma, dbeta);
  if (zero_out && hipMemsetAsync(zero_out, 0, (size_t)N * HW * C * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return CPM_ELAUNCH;
  return cpm::check_launch("groupnorm_backward");
}

CPM_EXPORT int cpm_groupnorm_backward(const float* dy, const float* x, const float* y, const float* gamma,
                                      const float* mean, const float* rstd, int N, int HW, int C, int G, int relu,
                                      float* dx, float* dgamma, float* dbeta, void* stream) {
  return groupnorm_backward_impl(dy, x, y, gamma, mean, rstd, N, HW, C, G, relu, dx, dgamma, dbeta, nullptr, stream);
}

CPM_EXPORT int cpm_groupnorm_backward_zero(const float* dy, const float* x, const float* y, const float* gamma,
                                           const float* mean, const float* rstd, int N, int HW, int C, int G, int relu,
                                           float* dx, float* dgamma, float* dbeta, float* zero_out, void* stream) {
  CPM_REQUIRE(zero_out, "null tensor to clear");
  return groupnorm_backward_impl(dy, x, y, gamma, mean, rstd, N, HW, C, G, relu, dx, dgamma, dbeta, zero_out, stream);
}

CPM_EXPORT int cpm_upsample2x_add_backward(const float* dy, int N, int P, int Q, int C, float* dtop, int accumulate,
                                           void* stream) {
  CPM_REQUIRE(dy && dtop, "null pointer");
  CPM_REQUIRE(C % 4 == 0, "C must be a multiple of 4");
  const int64_t total = (int64_t)N * ((P + 1) / 2) * ((Q + 1) / 2) * (C / 4);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, N, P, Q, C,
                     accumulate, dtop);
  return cpm::check_launch("upsample2x_add_backward");
}

static int sgd_step_impl(float* params, float* grads, float* momentum_buf, const int32_t* block_seg,
                         const int64_t* seg_end, const int32_t* seg_group, const float* h_group_lr,
                         const float* h_group_wd, int ngroups, int64_t begin, int64_t end, float momentum,
                         float grad_scale, int first_step, void* w4_out, void* stream, int zero_g = 0) {
  CPM_REQUIRE(params && grads && momentum_buf && block_seg && seg_end && seg_group && h_group_lr && h_group_wd,
              "null pointer");
  CPM_REQUIRE(ngroups > 0 && ngroups <= 8 && begin >= 0 && end >= begin && end % 64 == 0 && begin % 64 == 0,
              "bad shape (a range on 64-element boundaries, <= 8 groups)");
  CPM_REQUIRE(((uintptr_t)w4_out & 15) == 0, "unaligned image buffer");
  if (end == begin) return CPM_OK;
  SgdGroups grp = {};
  for (int i = 0; i < ngroups; ++i) { grp.lr[i] = h_group_lr[i]; grp.wd[i] = h_group_wd[i]; }
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for((end - begin) / 4, 256, 16384)), dim3(256), 0, (hipStream_t)stream, params,
                     grads, momentum_buf, block_seg, seg_end, seg_group, grp, end, momentum, grad_scale, first_step,
                     (uint4*)w4_out, begin, zero_g);
  return cpm::check_launch("sgd_step");
}

CPM_EXPORT int cpm_sgd_step(float* params, const float* grads, float* momentum_buf, const int32_t* block_seg,
                            const int64_t* seg_end, const int32_t* seg_group, const float* h_group_lr,
                            const float* h_group_wd, int ngroups, int64_t total, float momentum, float grad_scale,
                            int first_step, void* stream) {
  CPM_REQUIRE(total > 0, "bad shape (total % 64, <= 8 groups)");
  return sgd_step_impl(params, const_cast<float*>(grads), momentum_buf, block_seg, seg_end, seg_group, h_group_lr, h_group_wd,
                       ngroups, 0, total, momentum, grad_scale, first_step, nullptr, stream);
}

// the same, and the pre-split image (cpm_split_w4's format, `total` floats' worth of bytes at the parameters' own
// offsets) of the UPDATED parameters written by the same pass: the weight operand of the next step's forward convs
CPM_EXPORT int cpm_sgd_step_w4(float* params, const float* grads, float* momentum_buf, const int32_t* block_seg,
                               const int64_t* seg_end, const int32_t* seg_group, const float* h_group_lr,
                               const float* h_group_wd, int ngroups, int64_t total, float momentum, float grad_scale,
                               int first_step, void* w4_out, void* stream) {
  CPM_REQUIRE(w4_out && total > 0, "null image buffer");
  return sgd_step_impl(params, const_cast<float*>(grads), momentum_buf, block_seg, seg_end, seg_group, h_group_lr, h_group_wd,
                       ngroups, 0, total, momentum, grad_scale, first_step, w4_out, stream);
}

// the update of elements [begin, end) only (both on 64-element boundaries; pointers are those of the WHOLE buffers):
// a data-parallel trainer updates a chunk of the flat buffer as soon as its gradients are complete (and all-reduced),
// beside the rest of the backward pass.  w4_out may be NULL.  zero_grads != 0: every gradient element is set to zero
// behind its use -- the next step's zero_grad without its own pass over the buffer.
CPM_EXPORT int cpm_sgd_step_range(float* params, float* grads, float* momentum_buf, const int32_t* block_seg,
                                  const int64_t* seg_end, const int32_t* seg_group, const float* h_group_lr,
                                  const float* h_group_wd, int ngroups, int64_t begin, int64_t end, float momentum,
                                  float grad_scale, int first_step, void* w4_out, int zero_grads, void* stream) {
  return sgd_step_impl(params, grads, momentum_buf, block_seg, seg_end, seg_group, h_group_lr, h_group_wd, ngroups, begin,
                       end, momentum, grad_scale, first_step, w4_out, stream, zero_grads);
}

CPM_EXPORT int cpm_rpn_pred_backward_data(const float* const* dy_cls, const float* const* dy_box, const float* const* t,
                                          float* const* dx, const int64_t* pixels, int n_levels, const float* w_cls,
                                          const float* w_box, int A, int C, int gate, void* stream) {
  CPM_REQUIRE(dy_cls && dy_box && dx && pixels && w_cls && w_box && (t || !gate), "null pointer");
  CPM_REQUIRE(n_levels >= 1 && n_levels <= 8 && A >= 1 && C >= 4 && C % 4 == 0 && 256 % (C / 4) == 0,
              "1..8 levels, C a multiple of 4 with C / 4 dividing 256");
  CPM_REQUIRE((size_t)5 * A * C * sizeof(float) <= 64 * 1024, "5 A C floats of weights must fit 64 KB of LDS");
  PredLevels L = {};
  L.n = n_levels;
  int64_t tot = 0;
  for (int i = 0; i < n_levels; ++i) {
    CPM_REQUIRE(pixels[i] >= 0 && dy_cls[i] && dy_box[i] && dx[i] && (!gate || t[i]), "null level");
    CPM_REQUIRE((((uintptr_t)dx[i] | (uintptr_t)(gate ? t[i] : dx[i])) & 15) == 0, "16-byte aligned maps");
    L.dy_cls[i] = dy_cls[i]; L.dy_box[i] = dy_box[i]; L.t[i] = gate ? t[i] : nullptr; L.dx[i] = dx[i];
    tot += pixels[i];
    L.pix_end[i] = tot;
  }
  if (tot == 0) return CPM_OK;
  const int ppb = 256 / (C / 4);
  hipLaunchKernelGGL(rpn_pred_dgrad_kernel, dim3(grid_for(cpm::cdiv(tot, (int64_t)ppb) * 256, 256, 8192)), dim3(256),
                     (size_t)5 * A * C * sizeof(float), (hipStream_t)stream, L, w_cls, w_box, A, C, gate);
  return cpm::check_launch("rpn_pred_backward_data");
}
