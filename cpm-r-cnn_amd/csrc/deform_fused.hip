// Deformable (v1) / narrow-group 3x3 convolution with the sampling INSIDE the contraction: forward, data + offset
// gradient and weight gradient of ResNeXt-64x4d's grouped 3x3s (4..32 channels per group, BASELINE config #5), NHWC,
// gfx950.  No column matrix exists in memory on this path.
//
// Semantics: pet/lib/ops/csrc/Deformable/deform_conv_cuda_kernel.cu:95-128 (bilinear sampling), :215-287 (columns),
// :290-362 (input gradient), :365-460 (offset gradient); the contraction of deform_conv_cuda.cu:402-407 / 509-515 /
// 690-697 (forward, input gradient, weight gradient as per-group GEMMs over the column matrix).
//
// Design.  The unfused path (deform.hip) writes 9*C floats of columns per output pixel, contracts them with the grouped
// 1x1 MFMA kernels and scatters the column gradient back: 155 MB written and re-read several times per layer for a
// 17 MB map, and every bilinear corner fetched from L2.  Here a workgroup owns an 8x8 patch of OUTPUT pixels and a slab
// of 64 channels (whole groups: the layers this path takes have as many output as input channels per group).  The
// input window the patch's samples can reach -- the nominal footprint plus a margin of `mg` pixels -- is loaded into
// LDS ONCE (off-map cells are zero = the reference's "corner outside the map contributes 0"); a record per
// (pixel, tap) holds the sample's window cell and fractions.  Samples whose offsets leave the window read / update
// memory directly (rare: a trained offset predictor moves samples by a pixel or two).
//   A wave owns 16 channels of the slab.  The contraction runs on the f32 MFMA `v_mfma_f32_16x16x4_f32` (exact f32
// products and sums, MI355X_MICROARCH.md: 155 TFLOP/s measured; a layer has 1.24 GFLOP) with the layer's weights in
// REGISTERS as dense 16x16 blocks per tap -- block-diagonal for groups narrower than 16 channels (zeros cost MFMA time
// only, and these kernels are bound by LDS traffic and latency, not by the MFMA) and two blocks per tap for 32-channel
// groups.  A lane samples FOUR consecutive channels of ONE pixel per tap with four ds_read_b128 (the corners), which is
// the A fragment of four MFMAs: pixel = lane & 15, channels 4*(lane >> 4) .. +3.
//   forward (8 waves: slice x half of the patch):
//                    D[px][k] += sample[px][c] * W[c][k]                   -> affine / ReLU epilogue, y
//   data gradient (4 waves, one per slice, so that a wave OWNS its channels of the dx window):
//                    D[c][px]  = W[c][k] * dpre[k][px] (column gradient of one tap, in registers), then
//                    dx window in LDS += corner weight * D by plain read-add-write, the four corners of a tap in one
//                    round over 16 pixels with disjoint footprints (see the kernel), flushed with one float atomic per
//                    touched window element; samples outside the window
//                    add to memory directly, re-dealt so that an atomic instruction carries 64 contiguous bytes
//   weight + offset gradient (4 waves; both need every sample's corners from the x window):
//                    D[k][c] += dpre[k][px] * sample[px][c]: the sampled tile goes through a per-wave LDS buffer to
//                    change lanes from (pixel, 4 channels) to (channel, 4 pixels); workgroups walk several patches
//                    and add their 16x16 blocks to dw once.  d offset += column gradient * d(bilinear)/d(h|w), summed
//                    over the wave's channels by an MFMA against ones, over the slab's slices in LDS, over the slabs by
//                    one float atomic per (pixel, tap, h|w)
// Measured (tools/time_deform.py, layer3's 1024 x 50 x 84, offsets within +-1.5 px / +-20 px): forward 46 / 85 us
// (columns + grouped GEMM: 113), data gradient 173 / 364 (162 at +-0.3 px; GEMM + col2im: 291 / 444), weight + offset gradient 139 / 186
// (coord + weight GEMM: 146; but that path also needs the 155 MB of columns kept from the forward pass).
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DF_T = 8;                  // patch edge, output pixels
constexpr int DF_PX = DF_T * DF_T;       // 64 pixels = four 16-pixel MFMA tiles
constexpr int DF_SLAB = 64;              // channels per workgroup
constexpr int DF_PITCH = 68;             // floats per window cell (64 channels + 4: rows of a ds_read_b128 on distinct banks)
constexpr int DF_TAPS = 9;
constexpr int DF_THREADS = 512;          // 8 waves: (16-channel slice, half of the patch)

struct DfGeom {
  int N, H, W, C, P, Q, stride, pad, dil, cg_shift, dg;
  int win_h, win_w, mg, tiles_p, tiles_q;
};

struct DfPatch {
  int n, p0, q0, wy0, wx0;
};

__device__ __forceinline__ DfPatch df_patch(const DfGeom& G, int b) {
  DfPatch t;
  const int tq = b % G.tiles_q;
  b /= G.tiles_q;
  const int tp = b % G.tiles_p;
  t.n = b / G.tiles_p;
  t.p0 = tp * DF_T;
  t.q0 = tq * DF_T;
  t.wy0 = t.p0 * G.stride - G.pad - G.mg;
  t.wx0 = t.q0 * G.stride - G.pad - G.mg;
  return t;
}

// one record per (tap, pixel) of the patch: x = float index of the window cell of corner (h0, w0), -1 = a valid
// sample outside the window, -2 = nothing to sample (pixel beyond the map's edge, or the reference's validity test
// deform_conv_cuda_kernel.cu:262 fails); y / z = the fractions lh / lw; w = h0 and w0 (+2^15) packed
__device__ __forceinline__ void df_records(const DfGeom& G, const float* __restrict__ offset, const DfPatch& t, int dgu,
                                           int4* srec) {
  for (int idx = threadIdx.x; idx < DF_PX * DF_TAPS; idx += blockDim.x) {
    const int tap = idx >> 6, px = idx & 63;
    const int p = t.p0 + (px >> 3), q = t.q0 + (px & 7);
    int4 rec = make_int4(-2, 0, 0, 0);
    if (p < G.P && q < G.Q) {
      float oh = 0.f, ow = 0.f;
      if (offset) {
        const int64_t m = ((int64_t)t.n * G.P + p) * G.Q + q;
        const float2 o = *(const float2*)(offset + m * (2 * DF_TAPS * G.dg) + (dgu * DF_TAPS + tap) * 2);
        oh = o.x;
        ow = o.y;
      }
      const int i = tap / 3, j = tap - 3 * i;
      const float h = (float)(p * G.stride - G.pad + i * G.dil) + oh;
      const float w = (float)(q * G.stride - G.pad + j * G.dil) + ow;
      if (h > -1.f && w > -1.f && h < (float)G.H && w < (float)G.W) {
        const int h0 = (int)floorf(h), w0 = (int)floorf(w);
        const int wy = h0 - t.wy0, wx = w0 - t.wx0;
        const bool inside = wy >= 0 && wx >= 0 && wy + 1 < G.win_h && wx + 1 < G.win_w;
        rec.x = inside ? (wy * G.win_w + wx) * DF_PITCH : -1;
        rec.y = __float_as_int(h - (float)h0);
        rec.z = __float_as_int(w - (float)w0);
        rec.w = (int)(((unsigned)(h0 + 32768) << 16) | ((unsigned)(w0 + 32768) & 0xFFFFu));
      }
    }
    srec[idx] = rec;
  }
}

__device__ __forceinline__ void df_load_window(const DfGeom& G, const float* __restrict__ xb, int c0, const DfPatch& t,
                                               float* win) {
  const int cells = G.win_h * G.win_w;
  for (int idx = threadIdx.x; idx < cells * 16; idx += blockDim.x) {
    const int cell = idx >> 4, part = idx & 15;
    const int cy = cell / G.win_w;
    const int y = t.wy0 + cy, x = t.wx0 + (cell - cy * G.win_w);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)G.H && (unsigned)x < (unsigned)G.W)
      v = *(const f32x4*)(xb + ((int64_t)y * G.W + x) * G.C + c0 + part * 4);
    *(f32x4*)(win + cell * DF_PITCH + part * 4) = v;
  }
}

__device__ __forceinline__ void df_unpack(const int4 rec, int& h0, int& w0) {
  h0 = (int)((unsigned)rec.w >> 16) - 32768;
  w0 = (int)((unsigned)rec.w & 0xFFFFu) - 32768;
}

// the four corners (four consecutive channels each) of one sample; false when there is nothing to sample
template <bool DEFORM>
__device__ __forceinline__ bool df_corners(const float* win, const int4 rec, int coff, const DfGeom& G,
                                           const float* __restrict__ xb, int cglob, f32x4& a, f32x4& b, f32x4& d,
                                           f32x4& e) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  a = b = d = e = z;
  if (rec.x == -2) return false;
  if (rec.x >= 0) {
    const float* s = win + rec.x + coff;
    a = *(const f32x4*)s;
    if (DEFORM) {
      b = *(const f32x4*)(s + DF_PITCH);
      d = *(const f32x4*)(s + G.win_w * DF_PITCH);
      e = *(const f32x4*)(s + (G.win_w + 1) * DF_PITCH);
    }
    return true;
  }
  int h0, w0;
  df_unpack(rec, h0, w0);
  const float* s = xb + ((int64_t)h0 * G.W + w0) * G.C + cglob;
  const bool y0 = (unsigned)h0 < (unsigned)G.H, y1 = (unsigned)(h0 + 1) < (unsigned)G.H;
  const bool x0 = (unsigned)w0 < (unsigned)G.W, x1 = (unsigned)(w0 + 1) < (unsigned)G.W;
  if (y0 && x0) a = *(const f32x4*)s;
  if (DEFORM) {
    if (y0 && x1) b = *(const f32x4*)(s + G.C);
    if (y1 && x0) d = *(const f32x4*)(s + (int64_t)G.W * G.C);
    if (y1 && x1) e = *(const f32x4*)(s + (int64_t)(G.W + 1) * G.C);
  }
  return true;
}

template <bool DEFORM>
__device__ __forceinline__ f32x4 df_sample(const float* win, const int4 rec, int coff, const DfGeom& G,
                                           const float* __restrict__ xb, int cglob) {
  f32x4 a, b, d, e;
  if (!df_corners<DEFORM>(win, rec, coff, G, xb, cglob, a, b, d, e)) return a;
  if (!DEFORM) return a;                       // integer position: weight 1 on corner (h0, w0)
  const float lh = __int_as_float(rec.y), lw = __int_as_float(rec.z);
  const float hh = 1.f - lh, hw = 1.f - lw;
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * a + w2 * b + w3 * d + w4 * e;
}

// ---- forward ---------------------------------------------------------------------------------------------------------
template <int NS, bool DEFORM>
__global__ __launch_bounds__(DF_THREADS) void deform_fwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ offset,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int relu,
                                                                float* __restrict__ y, DfGeom G) {
  extern __shared__ __align__(16) float df_smem[];
  float* win = df_smem;
  int4* srec = (int4*)(df_smem + G.win_h * G.win_w * DF_PITCH);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = wave & 3, half = wave >> 2;
  const int l15 = lane & 15, lq = lane >> 4;
  const DfPatch t = df_patch(G, blockIdx.x);
  const int c0 = blockIdx.y * DF_SLAB;
  const float* xb = x + (int64_t)t.n * G.H * G.W * G.C;
  df_records(G, DEFORM ? offset : nullptr, t, c0 / (G.C / G.dg), srec);
  df_load_window(G, xb, c0, t, win);
  const int Cg = 1 << G.cg_shift;
  const int k0 = c0 + 16 * slice, k = k0 + l15;
  float breg[DF_TAPS][NS][4];
  int coff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c = (NS == 2 ? (k0 & ~31) + 16 * s : k0) + 4 * lq;
    coff[s] = c - c0;
    const bool same = (c >> G.cg_shift) == (k >> G.cg_shift);
    const float* wp = w + (int64_t)k * DF_TAPS * Cg + (c & (Cg - 1));
#pragma unroll
    for (int tap = 0; tap < DF_TAPS; ++tap) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (same) v = *(const f32x4*)(wp + tap * Cg);
      breg[tap][s][0] = v.x; breg[tap][s][1] = v.y; breg[tap][s][2] = v.z; breg[tap][s][3] = v.w;
    }
  }
  __syncthreads();
  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < DF_TAPS; ++tap) {
    int4 rec[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) rec[ti] = srec[tap * DF_PX + (2 * half + ti) * 16 + l15];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      f32x4 a[2];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) a[ti] = df_sample<DEFORM>(win, rec[ti], coff[s], G, xb, c0 + coff[s]);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
          acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti][kb], breg[tap][s][kb], acc[ti], 0, 0, 0);
      }
    }
  }
  const float sc = scale ? scale[k] : 1.f, sh = shift ? shift[k] : 0.f;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int px = (2 * half + ti) * 16 + 4 * lq + r;
      const int p = t.p0 + (px >> 3), q = t.q0 + (px & 7);
      if (p < G.P && q < G.Q) {
        float v = acc[ti][r];
        if (scale) v *= sc;
        if (shift) v += sh;
        if (relu) v = fmaxf(v, 0.f);
        y[(((int64_t)t.n * G.P + p) * G.Q + q) * G.C + k] = v;
      }
    }
  }
}

__device__ __forceinline__ void df_wave_fence() {
  // LDS operations of one wave execute in order; this keeps the COMPILER from moving them across the point
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int DF_BT = 256;               // threads of the backward kernels: one wave per 16-channel slice

// ---- data gradient ---------------------------------------------------------------------------------------------------
// LDS: [dx window] [records] [ranks].  Wave w owns channels [16w, 16w+16) of the slab in the window: nobody else
// touches them, and the LDS operations of one wave execute in order, so the window is updated with plain
// read-add-write of float4s, 16 pixels x 4 channel quads per instruction.  What that costs is the ROUND TRIP: a round
// is a dependent ds_read_b128 -> add -> ds_write_b128 (round 5's ablation: with one round per bilinear corner the 144
// rounds of a patch were 60 % of the kernel).  So the 16 pixels of an instruction are chosen such that their 2x2 corner
// footprints do not overlap -- the patch's four PARITY classes (y & 1, x & 1): same-class pixels are two apart, and a
// smooth offset field keeps their footprints apart -- and all four corners of a tap go in ONE round: four reads in
// flight, four adds, four writes (layer3 of X-101, offsets within +-0.3 px: 201 -> 160 us; the LDS unit is then active
// ~45 % of the kernel's cycles, profiles/round5_deform_pmc.txt).  Offsets that do bring two footprints of a group
// together are handled by ranks: a sample's rank is one more than the highest rank among the EARLIER samples of its
// (tap, class) group whose footprint touches its own (a greedy colouring, computed once per patch), and round j of a
// tap updates the samples of rank j: disjoint footprints within a round by construction, as many rounds as the deepest
// chain (one for a smooth field).  Integer positions (no offsets) touch one cell each, distinct by construction: rank 0.
// No LDS float atomics: ds_add_f32 was measured here at ~0.3 lane-adds per clock and CU -- 930 us for a layer when
// every update went that way.
template <bool DEFORM>
__device__ __forceinline__ int df_class_px(int cls, int l) {          // pixel id (y * 8 + x) of slot l of parity class cls
  if (!DEFORM) return cls * 16 + l;                                   // (no offsets: two rows of the patch, as the forward)
  return (2 * (l >> 2) + (cls >> 1)) * DF_T + 2 * (l & 3) + (cls & 1);
}

template <int NS, bool DEFORM>
__global__ __launch_bounds__(DF_BT, 2) void deform_bwd_dx_kernel(const float* __restrict__ dpre,
                                                                 const float* __restrict__ offset,
                                                                 const float* __restrict__ w, float* __restrict__ dx,
                                                                 DfGeom G) {
  extern __shared__ __align__(16) float df_smem[];
  const int cells = G.win_h * G.win_w;
  float* dwin = df_smem;
  int4* srec = (int4*)(dwin + cells * DF_PITCH);
  unsigned char* srank = (unsigned char*)(srec + DF_PX * DF_TAPS);
  float* tb = (float*)(srank + DF_PX * DF_TAPS) + (threadIdx.x >> 6) * 256;      // per wave: 16 pixels x 16 channels
  const int tid = threadIdx.x, lane = tid & 63, slice = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const DfPatch t = df_patch(G, blockIdx.x);
  const int c0 = blockIdx.y * DF_SLAB;
  float* dxb = dx + (int64_t)t.n * G.H * G.W * G.C;
  df_records(G, DEFORM ? offset : nullptr, t, c0 / (G.C / G.dg), srec);
  for (int i = tid; i < cells * (DF_PITCH / 4); i += DF_BT) ((f32x4*)dwin)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int Cg = 1 << G.cg_shift;
  const int cs0 = c0 + 16 * slice, c = cs0 + l15;              // A rows: this lane's input channel
  float areg[DF_TAPS][NS][4];
  int kq[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    kq[s] = (NS == 2 ? (cs0 & ~31) + 16 * s : cs0) + 4 * lq;                // four consecutive output channels
    const bool same = (kq[s] >> G.cg_shift) == (c >> G.cg_shift);
    const float* wp = w + (int64_t)kq[s] * DF_TAPS * Cg + (c & (Cg - 1));
#pragma unroll
    for (int tap = 0; tap < DF_TAPS; ++tap) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) areg[tap][s][kb] = same ? wp[((int64_t)kb * DF_TAPS + tap) * Cg] : 0.f;
    }
  }
  __syncthreads();
  // ranks: 16 lanes per (tap, class) group, sixteen groups at a time; pivot k hands its footprint and (final) rank to
  // the later samples of its group
  for (int grp = tid >> 4; grp < DF_TAPS * 4; grp += DF_BT / 16) {
    const int tap = grp >> 2, cls = grp & 3, l = tid & 15;
    const int idx = tap * DF_PX + df_class_px<DEFORM>(cls, l);
    int rank = 0;
    if (DEFORM) {
      const int mine = srec[idx].x;
      const int cell = mine >= 0 ? mine / DF_PITCH : 0;
      const int cy = mine >= 0 ? cell / G.win_w : -100000, cx = cell - (cell / G.win_w) * G.win_w;
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        const int py = __shfl(cy, k, 16), px = __shfl(cx, k, 16), pr = __shfl(rank, k, 16);
        const int dy = cy - py, dx_ = cx - px;
        if (l > k && dy >= -1 && dy <= 1 && dx_ >= -1 && dx_ <= 1 && rank <= pr) rank = pr + 1;
      }
    }
    srank[idx] = (unsigned char)rank;
  }
  __syncthreads();
  const int coff = 16 * slice + 4 * lq;                         // the D rows of this lane: channels cs0 + 4*lq .. +3
  auto load_dpre = [&](int tile, f32x4 (&v)[NS]) {
    const int px = df_class_px<DEFORM>(tile & 3, l15);
    const int p = t.p0 + (px >> 3), q = t.q0 + (px & 7);
    const bool ok = tile < 4 && p < G.P && q < G.Q;
    const float* dp = dpre + (((int64_t)t.n * G.P + (ok ? p : 0)) * G.Q + (ok ? q : 0)) * G.C;
#pragma unroll
    for (int s = 0; s < NS; ++s) v[s] = ok ? *(const f32x4*)(dp + kq[s]) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  f32x4 bnext[NS];
  load_dpre(0, bnext);
  for (int tile = 0; tile < 4; ++tile) {
    f32x4 breg[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) breg[s] = bnext[s];
    load_dpre(tile + 1, bnext);                       // in flight while this tile is scattered
    // the nine taps' column gradients (independent MFMA chains) and records of this tile, before the window updates
    // (whose wave fences keep the compiler from looking ahead)
    f32x4 gs[DF_TAPS];
    int4 recs[DF_TAPS];
    int ranks[DF_TAPS];
    const int mypx = df_class_px<DEFORM>(tile, l15);                // tile = parity class
#pragma unroll
    for (int tap = 0; tap < DF_TAPS; ++tap) {
      recs[tap] = srec[tap * DF_PX + mypx];
      ranks[tap] = srank[tap * DF_PX + mypx];
      gs[tap] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int tap = 0; tap < DF_TAPS; ++tap)
          gs[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[tap][s][kb], breg[s][kb], gs[tap], 0, 0, 0);
      }
    }
#pragma unroll
    for (int tap = 0; tap < DF_TAPS; ++tap) {
      const f32x4 g = gs[tap];
      const int4 rec = recs[tap];
      const int rank = ranks[tap];
      const bool valid = rec.x != -2, inwin = rec.x >= 0;
      const float lh = __int_as_float(rec.y), lw = __int_as_float(rec.z);
      const float hh = 1.f - lh, hw = 1.f - lw;
      const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
      // all corners of the tap in one round per rank: the reads are in flight together; a zero corner weight adds
      // nothing.  (Skipping a corner that weighs zero in EVERY lane -- integer positions: three of the four -- by a
      // ballot per corner was measured: the branches keep the reads from being issued together, 160 -> 286 us.)
      {
        const bool mine = valid && inwin;
        float* const sc = dwin + (mine ? rec.x : 0) + coff;
        for (int j = 0; __builtin_amdgcn_ballot_w64(mine && rank >= j) != 0; ++j) {
          if (mine && rank == j) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < (DEFORM ? 4 : 1); ++k)
              v[k] = *(const f32x4*)(sc + ((k >> 1) * G.win_w + (k & 1)) * DF_PITCH);
#pragma unroll
            for (int k = 0; k < (DEFORM ? 4 : 1); ++k)
              if (cw[k] != 0.f) *(f32x4*)(sc + ((k >> 1) * G.win_w + (k & 1)) * DF_PITCH) = v[k] + cw[k] * g;
          }
          df_wave_fence();                     // the next round reads cells this one wrote (through other lanes)
        }
      }
      // samples that left the window: direct float atomics.  The lanes are re-dealt through LDS from (pixel, 4 channels)
      // to (4 pixels) x (16 channels), so that an atomic instruction carries 64 contiguous bytes per sample corner
      // instead of four scattered words of sixteen samples (a layer whose offsets are tens of pixels: 1.5 -> 0.37 ms,
      // the rate the column path's scatter kernel reaches on the same offsets)
      if (__builtin_amdgcn_ballot_w64(valid && !inwin) != 0) {
        *(f32x4*)(tb + l15 * 16 + 4 * lq) = g;
        df_wave_fence();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int4 fr = srec[tap * DF_PX + df_class_px<DEFORM>(tile, 4 * j + lq)];
          if (fr.x != -1) continue;
          const float val = tb[(4 * j + lq) * 16 + l15];
          const float flh = __int_as_float(fr.y), flw = __int_as_float(fr.z);
          const float fhh = 1.f - flh, fhw = 1.f - flw;
          const float fw[4] = {fhh * fhw, fhh * flw, flh * fhw, flh * flw};
          int h0, w0;
          df_unpack(fr, h0, w0);
#pragma unroll
          for (int k = 0; k < (DEFORM ? 4 : 1); ++k) {
            const int yy = h0 + (k >> 1), xx = w0 + (k & 1);
            if (fw[k] != 0.f && (unsigned)yy < (unsigned)G.H && (unsigned)xx < (unsigned)G.W)
              unsafeAtomicAdd(dxb + ((int64_t)yy * G.W + xx) * G.C + cs0 + l15, fw[k] * val);
          }
        }
        df_wave_fence();
      }
    }
  }
  __syncthreads();
  // flush: a wave adds whole cells (64 channels = 256 contiguous bytes per atomic instruction)
#pragma unroll 4
  for (int cell = tid >> 6; cell < cells; cell += DF_BT / 64) {
    const int cy = cell / G.win_w;
    const int yy = t.wy0 + cy, xx = t.wx0 + (cell - cy * G.win_w);
    if ((unsigned)yy >= (unsigned)G.H || (unsigned)xx >= (unsigned)G.W) continue;
    const float v = dwin[cell * DF_PITCH + lane];
    if (v != 0.f) unsafeAtomicAdd(dxb + ((int64_t)yy * G.W + xx) * G.C + c0 + lane, v);
  }
}

// ---- weight gradient and offset gradient -----------------------------------------------------------------------------
// Both need the four corners of every sample from the x window, so they share one kernel.  LDS: [x window] [records]
// [per-wave sample buffer: 16 pixels x 16*NS channels].  grid (chunks, slabs): a workgroup walks the patches chunk,
// chunk + gridDim.x, ... of its slab with the 16x16 blocks of dw (9 taps x NS) in registers and adds them to dw once.
//   weight gradient: the sampled tile goes through the wave's LDS buffer to change lanes from (pixel, 4 channels) to
//     (channel, 4 pixels) = the B fragment of D[k][c] += dpre[k][px] * sample[px][c]
//   offset gradient: the tap's column gradient D[c][px] = W[c][k] * dpre[k][px] (as in the data gradient) times
//     d(bilinear)/d(h|w) of the corners, summed over the lane's 4 channels, the wave's 4 channel quads (two
//     cross-lane steps) and, through one float atomic per (pixel, tap, slice), over the layer's channels
template <int NS, bool DEFORM, bool WANT_W, bool WANT_OFF>
__global__ __launch_bounds__(DF_BT, 2) void deform_bwd_par_kernel(const float* __restrict__ dpre,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ offset,
                                                                  const float* __restrict__ w, float* __restrict__ dw,
                                                                  float* __restrict__ doffset, DfGeom G, int patches) {
  constexpr int COLP = 16 * NS + 4;
  extern __shared__ __align__(16) float df_smem[];
  float* win = df_smem;
  int4* srec = (int4*)(df_smem + G.win_h * G.win_w * DF_PITCH);
  const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* cb = (float*)(srec + DF_PX * DF_TAPS) + slice * 16 * COLP;
  float* soff = (float*)(srec + DF_PX * DF_TAPS) + 4 * 16 * COLP;       // [slice][tap][16 pixels][h, w] of one tile
  const int l15 = lane & 15, lq = lane >> 4;
  const int c0 = blockIdx.y * DF_SLAB;
  const int dgu = c0 / (G.C / G.dg);
  const int k0 = c0 + 16 * slice;
  const int Cg = 1 << G.cg_shift;
  f32x4 acc[DF_TAPS][NS];
  float wreg[DF_TAPS][NS][4];
  int coff[NS], kq[NS];
  const int s_own = NS == 2 ? (slice & 1) : 0;             // which of the sampled 16-channel blocks is this wave's own
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int cbase = NS == 2 ? (k0 & ~31) + 16 * s : k0;
    coff[s] = cbase - c0 + 4 * lq;
    kq[s] = cbase + 4 * lq;
#pragma unroll
    for (int tap = 0; tap < DF_TAPS; ++tap) acc[tap][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (WANT_OFF) {
      const int c = k0 + l15;
      const bool same = (kq[s] >> G.cg_shift) == (c >> G.cg_shift);
      const float* wp = w + (int64_t)kq[s] * DF_TAPS * Cg + (c & (Cg - 1));
#pragma unroll
      for (int tap = 0; tap < DF_TAPS; ++tap) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) wreg[tap][s][kb] = same ? wp[((int64_t)kb * DF_TAPS + tap) * Cg] : 0.f;
      }
    }
  }
  for (int patch = blockIdx.x; patch < patches; patch += gridDim.x) {
    const DfPatch t = df_patch(G, patch);
    const float* xb = x + (int64_t)t.n * G.H * G.W * G.C;
    __syncthreads();                                       // the previous patch's window and records are done with
    df_records(G, DEFORM ? offset : nullptr, t, dgu, srec);
    df_load_window(G, xb, c0, t, win);
    __syncthreads();
    auto load_dpre = [&](int tile, float (&a)[4], f32x4 (&b)[NS]) {
      if (WANT_W) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const int px = tile * 16 + 4 * lq + kb;
          const int p = t.p0 + (px >> 3), q = t.q0 + (px & 7);
          a[kb] = (tile < 4 && p < G.P && q < G.Q) ? dpre[(((int64_t)t.n * G.P + p) * G.Q + q) * G.C + k0 + l15] : 0.f;
        }
      }
      if (WANT_OFF) {
        const int px = tile * 16 + l15;
        const int p = t.p0 + (px >> 3), q = t.q0 + (px & 7);
        const bool ok = tile < 4 && p < G.P && q < G.Q;
        const int64_t m = ((int64_t)t.n * G.P + (ok ? p : 0)) * G.Q + (ok ? q : 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) b[s] = ok ? *(const f32x4*)(dpre + m * G.C + kq[s]) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    float aw_next[4];
    f32x4 bd_next[NS];
    load_dpre(0, aw_next, bd_next);
    for (int tile = 0; tile < 4; ++tile) {
      float aw[4];                                         // dpre[pixel 4*lq + kb of the tile][k0 + l15]
      f32x4 bd[NS];                                        // dpre[pixel l15 of the tile][kq .. +3]
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) aw[kb] = aw_next[kb];
#pragma unroll
      for (int s = 0; s < NS; ++s) bd[s] = bd_next[s];
      load_dpre(tile + 1, aw_next, bd_next);               // in flight while this tile is worked on
      const int px = tile * 16 + l15;
      const int p = t.p0 + (px >> 3), q = t.q0 + (px & 7);
      const bool ok = p < G.P && q < G.Q;
      const int64_t m = ((int64_t)t.n * G.P + (ok ? p : 0)) * G.Q + (ok ? q : 0);
      // the tile's records up front (the wave fences below keep the compiler from looking ahead) where registers allow
      constexpr bool PRE = !(NS == 2 && WANT_W && WANT_OFF);
      int4 recs[DF_TAPS];
      if (PRE) {
#pragma unroll
        for (int tap = 0; tap < DF_TAPS; ++tap) recs[tap] = srec[tap * DF_PX + px];
      }
      // one 16-channel block per tap (NS == 1): the NEXT tap's corners are requested before this tap's wave fences, so
      // that their latency -- LDS for samples in the window, memory for the others: ~1.5 us each, 36 in a row per patch
      // otherwise -- runs under this tap's work
      constexpr bool PIPE = NS == 1;
      f32x4 pa, pb, pd, pe;
      bool pany = false;
      if (PIPE) pany = df_corners<DEFORM>(win, recs[0], coff[0], G, xb, c0 + coff[0], pa, pb, pd, pe);
#pragma unroll
      for (int tap = 0; tap < DF_TAPS; ++tap) {
        const int4 rec = PRE ? recs[tap] : srec[tap * DF_PX + px];
        const float lh = __int_as_float(rec.y), lw = __int_as_float(rec.z);
        const float hh = 1.f - lh, hw = 1.f - lw;
        f32x4 na, nb, nd, ne;
        bool nany = false;
        if (PIPE && tap + 1 < DF_TAPS)
          nany = df_corners<DEFORM>(win, recs[tap + 1], coff[0], G, xb, c0 + coff[0], na, nb, nd, ne);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (WANT_OFF) {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
              g = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][s][kb], bd[s][kb], g, 0, 0, 0);
          }
        }
        float gh = 0.f, gw = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (!WANT_W && s != s_own) continue;
          f32x4 a, b, d, e;
          bool any;
          if (PIPE) {
            a = pa; b = pb; d = pd; e = pe;
            any = pany;
          } else {
            any = df_corners<DEFORM>(win, rec, coff[s], G, xb, c0 + coff[s], a, b, d, e);
          }
          if (WANT_W) {
            f32x4 v = a;
            if (DEFORM) v = (hh * hw) * a + (hh * lw) * b + (lh * hw) * d + (lh * lw) * e;
            *(f32x4*)(cb + l15 * COLP + s * 16 + 4 * lq) = v;
          }
          if (WANT_OFF && s == s_own && any) {
            // deform_conv_cuda_kernel.cu:185-209: d/dh = -(hw a + lw b) + (hw d + lw e), d/dw = -(hh a) + hh b - lh d + lh e
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              gh += g[r] * (-(hw * a[r]) - lw * b[r] + hw * d[r] + lw * e[r]);
              gw += g[r] * (-(hh * a[r]) + hh * b[r] - lh * d[r] + lh * e[r]);
            }
          }
        }
        if (WANT_OFF) {
          // the sum over the wave's four channel quads on the (idle) MFMA: D[px][j] = sum_quad partial[px][quad] * 1;
          // lane (j, quad q) then holds the sums of pixels 4q .. 4q+3
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          const f32x4 rh = __builtin_amdgcn_mfma_f32_16x16x4f32(gh, 1.f, z, 0, 0, 0);
          const f32x4 rw = __builtin_amdgcn_mfma_f32_16x16x4f32(gw, 1.f, z, 0, 0, 0);
          if (l15 < 4) {
            const float vh = l15 == 0 ? rh[0] : l15 == 1 ? rh[1] : l15 == 2 ? rh[2] : rh[3];
            const float vw = l15 == 0 ? rw[0] : l15 == 1 ? rw[1] : l15 == 2 ? rw[2] : rw[3];
            *(float2*)(soff + ((slice * DF_TAPS + tap) * 16 + 4 * lq + l15) * 2) = make_float2(vh, vw);
          }
        }
        if (WANT_W) {
          df_wave_fence();
#pragma unroll
          for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
              const float b = cb[(4 * lq + kb) * COLP + s * 16 + l15];
              acc[tap][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[kb], b, acc[tap][s], 0, 0, 0);
            }
          }
          df_wave_fence();
        }
        if (PIPE) {
          pa = na; pb = nb; pd = nd; pe = ne;
          pany = nany;
        }
      }
      if (WANT_OFF) {
        // the four slices' sums of this tile -> one float atomic per (pixel, tap, h|w) and slab
        __syncthreads();
        for (int i = threadIdx.x; i < DF_TAPS * 16 * 2; i += DF_BT) {
          const float v = soff[i] + soff[DF_TAPS * 32 + i] + soff[2 * DF_TAPS * 32 + i] + soff[3 * DF_TAPS * 32 + i];
          const int tp = i >> 5, pj = (i >> 1) & 15, px2 = tile * 16 + pj;
          const int p2 = t.p0 + (px2 >> 3), q2 = t.q0 + (px2 & 7);
          if (v != 0.f && p2 < G.P && q2 < G.Q)
            unsafeAtomicAdd(doffset + (((int64_t)t.n * G.P + p2) * G.Q + q2) * (2 * DF_TAPS * G.dg) +
                                (dgu * DF_TAPS + tp) * 2 + (i & 1), v);
        }
        __syncthreads();
      }
    }
  }
  if (WANT_W) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int c = (NS == 2 ? (k0 & ~31) + 16 * s : k0) + l15;        // D column: input channel
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + 4 * lq + r;                                 // D row: output channel
        if ((c >> G.cg_shift) != (k >> G.cg_shift)) continue;
        float* o = dw + (int64_t)k * DF_TAPS * Cg + (c & (Cg - 1));
#pragma unroll
        for (int tap = 0; tap < DF_TAPS; ++tap) unsafeAtomicAdd(o + tap * Cg, acc[tap][s][r]);
      }
    }
  }
}

enum { DF_FWD = 0, DF_DX = 1, DF_PAR = 2 };

struct DfPlan {
  DfGeom G[3];             // per kernel: the window margin differs with what else the kernel keeps in LDS
  size_t lds[3];
  int ns;
};

constexpr size_t DF_LDS_MAX = 160 * 1024, DF_LDS_TWO = 80 * 1024;      // one / two workgroups per CU

// 0 when this path takes the layer; the reason otherwise
const char* df_plan(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, int groups, int dg,
                    int P, int Q, DfPlan* out, bool deform = true) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || stride <= 0 || dil <= 0 || groups <= 0 || dg <= 0) return "bad geometry";
  if (R != 3 || S != 3) return "not a 3x3";
  if (K != C || C % groups) return "output channels differ from input channels";
  const int Cg = C / groups;
  if (Cg != 4 && Cg != 8 && Cg != 16 && Cg != 32) return "channels per group not in {4, 8, 16, 32}";
  if (C % DF_SLAB || C % dg || (C / dg) % DF_SLAB) return "a 64-channel slab crosses a deformable group";
  if (P != (H + 2 * pad - dil * 2 - 1) / stride + 1 || Q != (W + 2 * pad - dil * 2 - 1) / stride + 1) return "bad output size";
  if (H >= 32768 || W >= 32768) return "map too large";
  if ((int64_t)N * cpm::cdiv(P, DF_T) * cpm::cdiv(Q, DF_T) > 0x7fffffffll) return "too many patches";
  DfPlan p;
  p.ns = Cg == 32 ? 2 : 1;
  DfGeom G;
  G.N = N; G.H = H; G.W = W; G.C = C; G.P = P; G.Q = Q; G.stride = stride; G.pad = pad; G.dil = dil; G.dg = dg;
  G.cg_shift = Cg == 4 ? 2 : Cg == 8 ? 3 : Cg == 16 ? 4 : 5;
  G.tiles_p = cpm::cdiv(P, DF_T);
  G.tiles_q = cpm::cdiv(Q, DF_T);
  const size_t rec = (size_t)DF_PX * DF_TAPS * sizeof(int4);
  const size_t extra[3] = {0, (size_t)DF_PX * DF_TAPS + 4 * 256 * sizeof(float),
                           (size_t)4 * 16 * (16 * p.ns + 4) * sizeof(float) + (size_t)4 * DF_TAPS * 32 * sizeof(float)};
  for (int kind = 0; kind < 3; ++kind) {
    // margin 2 when two workgroups per CU still fit (stride 1: 15 x 15 cells = 61 KB), else margin 1 if THAT gets two,
    // else the widest margin one workgroup can hold (stride 2: 22 x 22 cells = 132 KB)
    // (no offsets -- the plain narrow-group 3x3 -- : every sample sits on the footprint, no margin: 11 x 11 cells = 33 KB,
    // three workgroups per CU)
    int best = deform ? -1 : 0;
    for (int pass = 0; pass < 2 && best < 0; ++pass) {
      for (int mg = 2; mg >= (pass == 0 ? 1 : 0); --mg) {
        const int win = (DF_T - 1) * stride + 2 * dil + 2 + 2 * mg;
        const size_t lds = (size_t)win * win * DF_PITCH * sizeof(float) + rec + extra[kind];
        if (lds <= (pass == 0 ? DF_LDS_TWO : DF_LDS_MAX)) {
          best = mg;
          break;
        }
      }
    }
    if (best < 0) return "the window does not fit in LDS";
    G.mg = best;
    G.win_h = G.win_w = (DF_T - 1) * stride + 2 * dil + 2 + 2 * best;
    p.G[kind] = G;
    p.lds[kind] = (size_t)G.win_h * G.win_w * DF_PITCH * sizeof(float) + rec + extra[kind];
  }
  *out = p;
  return nullptr;
}

int df_enabled() {
  static const int on = [] { const char* v = getenv("CPM_DEFORM_FUSED"); return v ? atoi(v) : 1; }();
  return on;
}

template <typename Kern>
int df_prepare(Kern kern, size_t lds) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      cpm::set_error("deform_conv (fused): hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
      return CPM_ELAUNCH;
    }
  }
  return CPM_OK;
}

}  // namespace

CPM_EXPORT int cpm_deform_conv_fused_supported(int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                               int dilation, int groups, int deformable_groups, int P, int Q) {
  DfPlan p;
  if (!df_enabled()) return 0;
  return df_plan(N, H, W, C, K, R, S, stride, pad, dilation, groups, deformable_groups, P, Q, &p) == nullptr ? 1 : 0;
}

#define DF_GO(KERN, THREADS, GRID, LDS, ...)                                                           \
  do {                                                                                                  \
    auto kern = KERN;                                                                                   \
    int rc = df_prepare(kern, LDS);                                                                     \
    if (rc != CPM_OK) return rc;                                                                        \
    hipLaunchKernelGGL(kern, GRID, dim3(THREADS), LDS, (hipStream_t)stream, __VA_ARGS__);               \
  } while (0)

CPM_EXPORT int cpm_deform_conv_forward(const float* x, const float* offset, const float* w, const float* scale,
                                       const float* shift, int relu, int N, int H, int W, int C, int K, int R, int S,
                                       int stride, int pad, int dilation, int groups, int deformable_groups, int P,
                                       int Q, float* y, void* stream) {
  DfPlan p;
  const char* why = df_plan(N, H, W, C, K, R, S, stride, pad, dilation, groups, deformable_groups, P, Q, &p, offset != nullptr);
  CPM_REQUIRE(why == nullptr, why);
  CPM_REQUIRE(x && w && y, "null pointer");
  const dim3 grid((unsigned)(N * p.G[DF_FWD].tiles_p * p.G[DF_FWD].tiles_q), (unsigned)(C / DF_SLAB));
#define DF_FWDK(NSV, DV) \
  DF_GO((deform_fwd_kernel<NSV, DV>), DF_THREADS, grid, p.lds[DF_FWD], x, offset, w, scale, shift, relu, y, p.G[DF_FWD])
  if (p.ns == 2) { if (offset) DF_FWDK(2, true); else DF_FWDK(2, false); }
  else { if (offset) DF_FWDK(1, true); else DF_FWDK(1, false); }
#undef DF_FWDK
  return cpm::check_launch("deform_conv_forward");
}

CPM_EXPORT int cpm_deform_conv_backward_data(const float* dpre, const float* offset, const float* w, int N, int H,
                                             int W, int C, int K, int R, int S, int stride, int pad, int dilation,
                                             int groups, int deformable_groups, int P, int Q, float* dx,
                                             void* stream) {
  DfPlan p;
  const char* why = df_plan(N, H, W, C, K, R, S, stride, pad, dilation, groups, deformable_groups, P, Q, &p, offset != nullptr);
  CPM_REQUIRE(why == nullptr, why);
  CPM_REQUIRE(dpre && w && dx, "null pointer");
  const dim3 grid((unsigned)(N * p.G[DF_DX].tiles_p * p.G[DF_DX].tiles_q), (unsigned)(C / DF_SLAB));
#define DF_DXK(NSV, DV) DF_GO((deform_bwd_dx_kernel<NSV, DV>), DF_BT, grid, p.lds[DF_DX], dpre, offset, w, dx, p.G[DF_DX])
  if (p.ns == 2) { if (offset) DF_DXK(2, true); else DF_DXK(2, false); }
  else { if (offset) DF_DXK(1, true); else DF_DXK(1, false); }
#undef DF_DXK
  return cpm::check_launch("deform_conv_backward_data");
}

CPM_EXPORT int cpm_deform_conv_backward_params(const float* dpre, const float* x, const float* offset, const float* w,
                                               int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                               int dilation, int groups, int deformable_groups, int P, int Q,
                                               float* dw, float* doffset, void* stream) {
  DfPlan p;
  const char* why = df_plan(N, H, W, C, K, R, S, stride, pad, dilation, groups, deformable_groups, P, Q, &p, offset != nullptr);
  CPM_REQUIRE(why == nullptr, why);
  CPM_REQUIRE(dpre && x, "null pointer");
  CPM_REQUIRE(dw || doffset, "nothing asked for");
  CPM_REQUIRE(!doffset || (offset && w), "an offset gradient needs the offsets and the weight");
  if (doffset) {
    const size_t bytes = (size_t)N * P * Q * 2 * DF_TAPS * deformable_groups * sizeof(float);
    hipError_t e = hipMemsetAsync(doffset, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
      cpm::set_error("deform_conv_backward_params: memset: %s", hipGetErrorString(e));
      return CPM_ELAUNCH;
    }
  }
  const int patches = N * p.G[DF_PAR].tiles_p * p.G[DF_PAR].tiles_q, slabs = C / DF_SLAB;
  // a weight gradient: about four workgroups per CU across the slabs (two resident), each adding its 64 x 9 x Cg
  // block to dw once at its end; an offset gradient alone: a workgroup per patch
  int chunks = dw ? cpm::cdiv(4 * 256, slabs) : patches;
  if (chunks > patches) chunks = patches;
  if (chunks < 1) chunks = 1;
  chunks = cpm::cdiv(patches, cpm::cdiv(patches, chunks));          // equal walks
  const dim3 grid((unsigned)chunks, (unsigned)slabs);
#define DF_PAR(NSV, DV, WV, OV) \
  DF_GO((deform_bwd_par_kernel<NSV, DV, WV, OV>), DF_BT, grid, p.lds[DF_PAR], dpre, x, offset, w, dw, doffset, p.G[DF_PAR],  \
        patches)
#define DF_PAR_NS(NSV)                                                   \
  do {                                                                   \
    if (!offset) DF_PAR(NSV, false, true, false);                        \
    else if (dw && doffset) DF_PAR(NSV, true, true, true);               \
    else if (dw) DF_PAR(NSV, true, true, false);                         \
    else DF_PAR(NSV, true, false, true);                                 \
  } while (0)
  CPM_REQUIRE(offset || dw, "an offset gradient needs the offsets");
  if (p.ns == 2) DF_PAR_NS(2); else DF_PAR_NS(1);
#undef DF_PAR_NS
#undef DF_PAR
  return cpm::check_launch("deform_conv_backward_params");
}
