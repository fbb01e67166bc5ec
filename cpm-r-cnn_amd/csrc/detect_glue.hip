// Detection "glue" of the CPM training step as fused device kernels: RoI<->gt matching, the grid
// heat-map loss with its targets rasterised on the fly, and the heat-map -> box decoder.  Each replaces a chain of
// 25..60 small framework launches per image and stage in the reference (and its .cpu()/.cuda() round trips) with
// ONE launch over all images of the batch.  HBM/latency-bound integer and index work: no MFMA here.
//
// Built with -ffp-contract=off: every threshold comparison and truncation below must see the same fp32 values
// as the reference's op-by-op tensor arithmetic.
//
//   match   pet/utils/data/structures/boxlist_ops.py:123-158 (IoU, "+1" widths) + pet/rcnn/utils/matcher.py:48-111
//   loss    pet/rcnn/modeling/grid_cascade_rcnn/loss.py:178-258 (targets) + :260-262 (BCE with logits, mean)
//   decode  pet/rcnn/modeling/grid_cascade_rcnn/inference.py:189-279 (get_boxes), :281-290 (_filter_boxes)
#include "common.h"

namespace {

constexpr int MAX_POINTS = 16;

__device__ __forceinline__ float iou_plus1(const float4 g, const float4 p) {
  // box_iou_plus1(gt, prediction): the gt is "box1"
  const float area1 = (g.z - g.x + 1.f) * (g.w - g.y + 1.f);
  const float area2 = (p.z - p.x + 1.f) * (p.w - p.y + 1.f);
  const float ltx = fmaxf(g.x, p.x), lty = fmaxf(g.y, p.y);
  const float rbx = fminf(g.z, p.z), rby = fminf(g.w, p.w);
  const float w = fmaxf(rbx - ltx + 1.f, 0.f), h = fmaxf(rby - lty + 1.f, 0.f);
  const float inter = w * h;
  return inter / (area1 + area2 - inter);
}

// per-gt maximum IoU over all predictions of its image (Matcher.set_low_quality_matches_, matcher.py:91-93);
// IoU >= 0, so the int view of the float orders like the float
constexpr int RM_ITEMS = 8;     // predictions per thread in the row-maximum pass

constexpr int RM_THREADS = 512, RM_WAVES = RM_THREADS / 64, RM_GTS = 128;

__global__ __launch_bounds__(RM_THREADS) void match_rowmax_kernel(const float4* __restrict__ rois,
                                                                  const int* __restrict__ roi_img,
                                                                  const float4* __restrict__ gts,
                                                                  const int* __restrict__ gt_off, int R,
                                                                  int* __restrict__ row_max) {
  // 268 k anchors per image against a few dozen gts whose maxima share ONE cache line: every atomic on it is served
  // alone (~4 ns), and a value read beforehand is no filter -- all workgroups start together, on zeros.  A thread
  // therefore folds RM_ITEMS predictions per gt in registers, a wave whose predictions all belong to one image
  // reduces across its 64 lanes, the workgroup's 16 waves meet in LDS, and ONE atomic per gt and workgroup (8 192
  // predictions) goes out.  A wave straddling an image boundary (or an image with more than RM_GTS gts) issues its
  // own atomic per gt.
  __shared__ float s_max[RM_WAVES][RM_GTS];
  __shared__ int s_img[RM_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = (blockIdx.x * RM_THREADS + (threadIdx.x & ~63)) * RM_ITEMS + lane;
  float4 p[RM_ITEMS];
  int img[RM_ITEMS];
  bool uniform = true;
  int img0 = -1;
#pragma unroll
  for (int k = 0; k < RM_ITEMS; ++k) {
    const int i = base + k * 64;
    const bool live = i < R;
    img[k] = live ? (roi_img ? roi_img[i] : 0) : -1;
    p[k] = live ? rois[i] : make_float4(0.f, 0.f, -2.f, -2.f);       // dead slot: empty box, IoU 0 with anything
    if (k == 0) img0 = __shfl(img[0], 0, 64);
    uniform = uniform && __all(img[k] == img0 || !live);
  }
  uniform = uniform && img0 >= 0;
  const bool pooled = uniform && gt_off[img0 + 1] - gt_off[img0] <= RM_GTS;   // this wave's maxima go through LDS
  if (lane == 0) s_img[wave] = pooled ? img0 : -1;
  // the images present in the wave, one after the other (one, except for the wave at an image boundary, whose
  // per-prediction atomics -- 8 192 turns on one cache line -- were most of this kernel's time)
  int done_img = -1;
  for (;;) {
    int im = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < RM_ITEMS; ++k)
      if (img[k] > done_img) im = min(im, img[k]);
    for (int o = 32; o > 0; o >>= 1) im = min(im, __shfl_xor(im, o, 64));
    if (im == 0x7fffffff) break;
    done_img = im;
    const int g0 = gt_off[im], ng = gt_off[im + 1] - g0;
    float4 gt_lane = make_float4(0.f, 0.f, 0.f, 0.f);               // gt (g & ~63) + lane: one load per 64 gts, not one per gt
    for (int g = 0; g < ng; ++g) {
      if ((g & 63) == 0) gt_lane = (g + lane < ng) ? gts[g0 + g + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 gt = make_float4(__shfl(gt_lane.x, g & 63, 64), __shfl(gt_lane.y, g & 63, 64),
                                    __shfl(gt_lane.z, g & 63, 64), __shfl(gt_lane.w, g & 63, 64));
      // 64 neighbouring anchors x 8 rarely touch a given gt at all: then every IoU is 0, and the wave skips the eight
      // divisions and the reduction (the overlap test is iou_plus1's own intersection, so "no overlap" is exactly v == 0)
      bool touch = false;
#pragma unroll
      for (int k = 0; k < RM_ITEMS; ++k) {
        const float w = fminf(gt.z, p[k].z) - fmaxf(gt.x, p[k].x) + 1.f, h = fminf(gt.w, p[k].w) - fmaxf(gt.y, p[k].y) + 1.f;
        touch = touch || (img[k] == im && w > 0.f && h > 0.f);
      }
      if (!__any(touch)) {
        if (lane == 0 && pooled) s_max[wave][g] = 0.f;
        continue;
      }
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < RM_ITEMS; ++k)
        if (img[k] == im) v = fmaxf(v, iou_plus1(gt, p[k]));
      for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
      if (lane == 0) {
        if (pooled) s_max[wave][g] = v;
        else if (v > 0.f) atomicMax(row_max + g0 + g, __float_as_int(v));
      }
    }
  }
  __syncthreads();
  // the pooled waves, image by image (a workgroup spans two images at most once per image boundary)
  for (int w0 = 0; w0 < RM_WAVES; ++w0) {
    const int im = s_img[w0];
    if (im < 0 || (w0 > 0 && s_img[w0 - 1] == im)) continue;          // first wave of a run of one image
    const int b0 = gt_off[im], n = gt_off[im + 1] - b0;
    for (int g = threadIdx.x; g < n; g += RM_THREADS) {
      float v = 0.f;
      for (int w = w0; w < RM_WAVES && s_img[w] == im; ++w) v = fmaxf(v, s_max[w][g]);
      if (v > 0.f) atomicMax(row_max + b0 + g, __float_as_int(v));
    }
  }
}

__global__ __launch_bounds__(256) void match_kernel(const float4* __restrict__ rois, const int* __restrict__ roi_img,
                                                    const float4* __restrict__ gts, const int* __restrict__ gt_off,
                                                    int R, float high, float low, const int* __restrict__ row_max,
                                                    int64_t* __restrict__ matched, float* __restrict__ max_iou) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const int img = roi_img ? roi_img[i] : 0;
  const float4 p = rois[i];
  const int g0 = gt_off[img], g1 = gt_off[img + 1];
  float best = -1.f;
  int arg = 0;
  bool tied = false;
  for (int g = g0; g < g1; ++g) {
    const float v = iou_plus1(gts[g], p);
    if (v > best) { best = v; arg = g - g0; }                  // first maximum wins
    if (row_max && v == __int_as_float(row_max[g])) tied = true;
  }
  int64_t m = arg;
  if (best < low) m = -1;                                       // BELOW_LOW_THRESHOLD
  else if (best < high) m = -2;                                 // BETWEEN_THRESHOLDS
  if (tied) m = arg;                                            // low-quality match restored
  matched[i] = m;
  if (max_iou) max_iou[i] = best;
}

struct GridGeom {
  int P, gs, map, half;             // points, sqrt(points), whole map (56), sub-region side (28)
  int sub_x[MAX_POINTS], sub_y[MAX_POINTS];
  float fx[MAX_POINTS], fy[MAX_POINTS];
  long sR, sC, sH, sW;              // element strides of the logits tensor [R, P, half, half]
};

// one wave per (RoI, point): BCE-with-logits against the point's rasterised target, summed into *loss_sum
// (already divided by the element count and multiplied by `weight`), gradient written alongside
__global__ __launch_bounds__(64) void grid_bce_kernel(const float* __restrict__ logits,
                                                      const float4* __restrict__ rois,
                                                      const float4* __restrict__ gt, GridGeom G, float ratio,
                                                      int radius, float scale, float* __restrict__ loss_sum,
                                                      float* __restrict__ grad) {
  const int r = blockIdx.x / G.P, j = blockIdx.x - r * G.P;
  const int lane = threadIdx.x;
  const float4 b = rois[r], g = gt[r];
  const float x1 = b.x - ratio * ((b.z - b.x) / 2.f), y1 = b.y - ratio * ((b.w - b.y) / 2.f);
  const float x2 = b.z + ratio * ((b.z - b.x) / 2.f), y2 = b.w + ratio * ((b.w - b.y) / 2.f);
  const float bw = x2 - x1, bh = y2 - y1;
  const bool ok = !(bw <= (float)G.gs || bh <= (float)G.gs);                    // "ignore small bboxes"
  int cx = 0, cy = 0;
  if (ok) {
    const float fx = G.fx[j], fy = G.fy[j];
    const float gx = fx * g.x + (1.f - fx) * g.z, gy = fy * g.y + (1.f - fy) * g.w;
    cx = (int)((gx - x1) / bw * (float)G.map);                                  // python int(): toward zero
    cy = (int)((gy - y1) / bh * (float)G.map);
  }
  const float* src = logits + r * G.sR + j * G.sC;
  float* dst = grad + r * G.sR + j * G.sC;
  float acc = 0.f;
  const int n = G.half * G.half;
  for (int k = lane; k < n; k += 64) {
    const int row = k / G.half, col = k - row * G.half;
    const int dx = col + G.sub_x[j] - cx, dy = row + G.sub_y[j] - cy;
    const float t = (ok && dx * dx + dy * dy <= radius * radius) ? 1.f : 0.f;
    const long off = row * G.sH + col * G.sW;
    const float x = src[off];
    const float e = expf(-fabsf(x));
    // (1 - t) * x - log_sigmoid(x),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
    acc += (1.f - t) * x - (fminf(x, 0.f) - log1pf(e));
    const float sig = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dst[off] = (sig - t) * scale;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) atomicAdd(loss_sum, acc * scale);
}

// one workgroup per RoI, one wave per point: first arg-max of sigmoid(logit) in the point's window, then the
// score-weighted vote of the gs points on each box side; optional "coincides with a gt coordinate" filter
__global__ __launch_bounds__(64 * MAX_POINTS) void grid_decode_kernel(
    const float* __restrict__ logits, const float4* __restrict__ rois, GridGeom G, float ratio,
    const int* __restrict__ roi_img, const float4* __restrict__ gts, const int* __restrict__ gt_off,
    float4* __restrict__ out, unsigned char* __restrict__ keep) {
  __shared__ float s_score[MAX_POINTS], s_ax[MAX_POINTS], s_ay[MAX_POINTS];
  const int r = blockIdx.x, j = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4 b = rois[r];
  const float wdt = b.z - b.x, hgt = b.w - b.y;
  const float x1 = b.x - ratio * (wdt / 2.f), y1 = b.y - ratio * (hgt / 2.f);
  if (j < G.P) {
    const float* src = logits + r * G.sR + j * G.sC;
    const int n = G.half * G.half;
    float best = -1.f;
    int bi = n;
    for (int k = lane; k < n; k += 64) {
      const int row = k / G.half, col = k - row * G.half;
      const float x = src[row * G.sH + col * G.sW];
      const float p = 1.f / (1.f + expf(-x));
      if (p > best) { best = p; bi = k; }                        // ascending k per lane: first maximum
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
      const int xs = bi % G.half + G.sub_x[j], ys = bi / G.half + G.sub_y[j];
      s_score[j] = best;
      s_ax[j] = ((float)xs + 0.5f) / (float)(2 * G.half) * (1.f + ratio) * wdt + x1;
      s_ay[j] = ((float)ys + 0.5f) / (float)(2 * G.half) * (1.f + ratio) * hgt + y1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float nx1 = 0, dx1 = 0, ny1 = 0, dy1 = 0, nx2 = 0, dx2 = 0, ny2 = 0, dy2 = 0;
    for (int k = 0; k < G.gs; ++k) {
      const int ix1 = k, iy1 = k * G.gs, ix2 = G.P - G.gs + k, iy2 = (k + 1) * G.gs - 1;
      nx1 += s_ax[ix1] * s_score[ix1]; dx1 += s_score[ix1];
      ny1 += s_ay[iy1] * s_score[iy1]; dy1 += s_score[iy1];
      nx2 += s_ax[ix2] * s_score[ix2]; dx2 += s_score[ix2];
      ny2 += s_ay[iy2] * s_score[iy2]; dy2 += s_score[iy2];
    }
    out[r] = make_float4(nx1 / dx1, ny1 / dy1, nx2 / dx2, ny2 / dy2);
    if (keep) {
      // inference.py:281-290: a coordinate equal to the same coordinate of ANY gt of the image becomes -1; the RoI
      // survives while ((c0 + c1) + c2) + c3 > 0
      const int img = roi_img ? roi_img[r] : 0;
      float c0 = b.x, c1 = b.y, c2 = b.z, c3 = b.w;
      for (int g = gt_off[img]; g < gt_off[img + 1]; ++g) {
        const float4 t = gts[g];
        if (b.x == t.x) c0 = -1.f;
        if (b.y == t.y) c1 = -1.f;
        if (b.z == t.z) c2 = -1.f;
        if (b.w == t.w) c3 = -1.f;
      }
      keep[r] = (((c0 + c1) + c2) + c3 > 0.f) ? 1 : 0;
    }
  }
}

// RPN candidates of one FPN level: gather the top-k regression rows and their anchors, BoxCoder.decode
// (box_coder.py:51-94: pixel-inclusive widths, dw/dh clipped before exp, x2/y2 get the -1 back) and clip_to_image
// (bounding_box.py:233-243, remove_empty=False) -- one launch per level instead of ~35 tensor ops
struct ImSizes { float w[64], h[64]; };

__global__ __launch_bounds__(256) void rpn_decode_kernel(const float4* __restrict__ reg, const int64_t* __restrict__ idx,
                                                         const float4* __restrict__ anchors, int N, int A, int k,
                                                         float wx, float wy, float ww, float wh, float clip,
                                                         ImSizes sz, float4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * k) return;
  const int n = i / k;
  const int64_t j = idx[i];
  const float4 b = anchors[j], c = reg[(int64_t)n * A + j];
  const float w = b.z - b.x + 1.f, h = b.w - b.y + 1.f;
  const float cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
  const float dx = c.x / wx, dy = c.y / wy;
  const float dw = fminf(c.z / ww, clip), dh = fminf(c.w / wh, clip);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  const float mx = sz.w[n] - 1.f, my = sz.h[n] - 1.f;
  float4 o;
  o.x = fminf(fmaxf(pcx - 0.5f * pw, 0.f), mx);
  o.y = fminf(fmaxf(pcy - 0.5f * ph, 0.f), my);
  o.z = fminf(fmaxf(pcx + 0.5f * pw - 1.f, 0.f), mx);
  o.w = fminf(fmaxf(pcy + 0.5f * ph - 1.f, 0.f), my);
  out[i] = o;
}

// all FPN levels in one launch (blockIdx.y = level), results written straight into the level-major segment layout
// the batched NMS reads (segment = level * N + image, k_l rows each)
constexpr int DEC_LEVELS = 8;
struct DecodeLevels {
  const float4* reg[DEC_LEVELS];
  const int64_t* idx[DEC_LEVELS];
  const float4* anchors[DEC_LEVELS];
  int A[DEC_LEVELS], k[DEC_LEVELS], out_off[DEC_LEVELS];
};

__global__ __launch_bounds__(256) void rpn_decode_multi_kernel(DecodeLevels lv, int N, float wx, float wy, float ww,
                                                               float wh, float clip, ImSizes sz,
                                                               float4* __restrict__ out) {
  const int l = blockIdx.y, k = lv.k[l], A = lv.A[l];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * k) return;
  const int n = i / k;
  const int64_t j = lv.idx[l][i];
  const float4 b = lv.anchors[l][j], c = lv.reg[l][(int64_t)n * A + j];
  const float w = b.z - b.x + 1.f, h = b.w - b.y + 1.f;
  const float cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
  const float dx = c.x / wx, dy = c.y / wy;
  const float dw = fminf(c.z / ww, clip), dh = fminf(c.w / wh, clip);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  const float mx = sz.w[n] - 1.f, my = sz.h[n] - 1.f;
  float4 o;
  o.x = fminf(fmaxf(pcx - 0.5f * pw, 0.f), mx);
  o.y = fminf(fmaxf(pcy - 0.5f * ph, 0.f), my);
  o.z = fminf(fmaxf(pcx + 0.5f * pw - 1.f, 0.f), mx);
  o.w = fminf(fmaxf(pcy + 0.5f * ph - 1.f, 0.f), my);
  out[lv.out_off[l] + i] = o;
}

// objectness logits of every level -> sigmoid, one launch (RPNPostProcessor: objectness.sigmoid(), inference.py:72);
// the expression torch's sigmoid kernel evaluates, so the scores -- and every tie among them -- are the same bits
struct SigLevels { const float* in[DEC_LEVELS]; int n[DEC_LEVELS], out_off[DEC_LEVELS]; };

__global__ __launch_bounds__(256) void sigmoid_multi_kernel(SigLevels lv, float* __restrict__ out) {
  const int l = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= lv.n[l]) return;
  out[lv.out_off[l] + i] = 1.f / (1.f + expf(-lv.in[l][i]));
}

// RPN anchor labels from the match (rpn/loss.py:60-79): 1 matched, 0 below the low threshold, -1 between the
// thresholds or not visible
__global__ __launch_bounds__(256) void rpn_labels_kernel(const int64_t* __restrict__ matched,
                                                         const uint8_t* __restrict__ vis, int64_t total,
                                                         int discard_between, float* __restrict__ lab) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t m = matched[i];
  float v = m >= 0 ? 1.f : 0.f;
  if (discard_between && m == -2) v = -1.f;
  if (vis && !vis[i]) v = -1.f;
  lab[i] = v;
}

// ISM loss: l2_loss (pet/lib/ops/l2_loss.py:4-11) of [R, 2] predictions against targets (1 - iou, iou).  The reference
// indexes x[pos_inds] with the [P, 2] (row, column) pairs of the positive targets, i.e. every positive entry (r, c)
// gathers rows r AND c; with E[i] = 0.5 * sum_j (x[i,j] - t[i,j])^2 that is
//   (sum_r cnt[r] * E[r] + n_col0 * E[0] + n_col1 * E[1]) / P,   cnt[r] = #positive targets of row r.
// Value and gradient in one launch of one workgroup (R is a few hundred).
__global__ __launch_bounds__(256) void l2_pairs_kernel(const float2* __restrict__ x, const float* __restrict__ iou,
                                                       const float2* __restrict__ target, int R,
                                                       float* __restrict__ loss, float2* __restrict__ grad) {
  __shared__ float s_red[256];
  __shared__ float s_col[2];
  auto tgt = [&](int r) { return target ? target[r] : make_float2(1.f - iou[r], iou[r]); };
  float c0 = 0.f, c1 = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float2 t = tgt(r);
    c0 += t.x > 0.f ? 1.f : 0.f;
    c1 += t.y > 0.f ? 1.f : 0.f;
  }
  for (int pass = 0; pass < 2; ++pass) {
    s_red[threadIdx.x] = pass ? c1 : c0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) s_col[pass] = s_red[0];
    __syncthreads();
  }
  const float col0 = s_col[0], col1 = s_col[1];
  const float P = col0 + col1, inv = 1.f / fmaxf(P, 1.f);
  float part = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float2 t = tgt(r), v = x[r];
    float w = (t.x > 0.f ? 1.f : 0.f) + (t.y > 0.f ? 1.f : 0.f);
    if (r == 0) w += col0;
    if (r == 1) w += col1;
    const float dx = v.x - t.x, dy = v.y - t.y;
    part += w * 0.5f * (dx * dx + dy * dy);
    grad[r] = make_float2(w * dx * inv, w * dy * inv);
  }
  s_red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = s_red[0] * inv;
}

// F.cross_entropy(logits [R, C], labels [R], reduction = mean, ignore_index) -- log_softmax + nll_loss -- as ONE launch
// for value and gradient (the cls and RSM heads' losses, grid_cascade_rcnn/loss.py:103-112): the framework formulation is
// two launches forward and two backward per head, all four queued where the device waits for the host (the end of the
// forward pass, the start of the backward pass).  A wave per row, lanes over the classes, four rows per workgroup
// (a row is one dependent chain of a load, two cross-lane reductions, an exp and a log: one workgroup walking 1024
// rows took 0.3 ms).  Every workgroup counts the valid rows itself (R labels from L2); the row losses go to
// `row_loss` and the workgroup that arrives last (ticket) sums them in a fixed order: the same bits every run.
// Rows whose label is ignore_index add nothing and get a zero gradient; the mean is over the others.
__device__ __forceinline__ float ce_wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float ce_wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float ce_block_sum(float v, float* s_red) {     // 256 threads, fixed order
  v = ce_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels,
                                                         int R, int C, int64_t ignore_index, float* __restrict__ loss,
                                                         float* __restrict__ grad, float* __restrict__ row_loss,
                                                         int* __restrict__ ticket) {
  __shared__ float s_red[4];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cnt = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) cnt += labels[r] != ignore_index ? 1.f : 0.f;
  const float n = ce_block_sum(cnt, s_red);
  const float inv = n > 0.f ? 1.f / n : 0.f;
  const int r = blockIdx.x * 4 + wave;
  if (r < R) {
    const int64_t lab = labels[r];
    const float* xr = x + (size_t)r * C;
    float* gr = grad + (size_t)r * C;
    if (lab == ignore_index) {                               // (wave-uniform)
      for (int c = lane; c < C; c += 64) gr[c] = 0.f;
      if (lane == 0) row_loss[r] = 0.f;
    } else {
      float m = -INFINITY;
      for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
      m = ce_wave_max(m);
      float e = 0.f;
      for (int c = lane; c < C; c += 64) e += expf(xr[c] - m);
      const float lse = m + logf(ce_wave_sum(e));
      for (int c = lane; c < C; c += 64) gr[c] = (expf(xr[c] - lse) - (c == (int)lab ? 1.f : 0.f)) * inv;
      if (lane == 0) row_loss[r] = lse - xr[lab];
    }
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  float part = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) part += __builtin_nontemporal_load(row_loss + i);
  const float t = ce_block_sum(part, s_red);
  if (threadIdx.x == 0) {
    *loss = t / n;                                           // no valid row: 0 / 0, as the framework's mean
    *ticket = 0;                                             // ready for the next call
  }
}

int fill_geom(GridGeom& G, int points, int map_size, const int* sub_xy, const int64_t* strides) {
  if (points <= 0 || points > MAX_POINTS) return -1;
  int gs = 1;
  while (gs * gs < points) ++gs;
  if (gs * gs != points || gs < 2) return -1;
  G.P = points; G.gs = gs; G.map = map_size; G.half = map_size / 4 * 2;
  for (int j = 0; j < points; ++j) {
    G.sub_x[j] = sub_xy[2 * j];
    G.sub_y[j] = sub_xy[2 * j + 1];
    if (G.sub_x[j] < 0 || G.sub_y[j] < 0 || G.sub_x[j] + G.half > map_size || G.sub_y[j] + G.half > map_size) return -1;
    G.fx[j] = (float)(1.0 - (double)(j / gs) / (gs - 1));       // loss.py:205-209
    G.fy[j] = (float)(1.0 - (double)(j % gs) / (gs - 1));
  }
  G.sR = strides[0]; G.sC = strides[1]; G.sH = strides[2]; G.sW = strides[3];
  return 0;
}


// ---- RPN loss: targets + both losses + both gradients in one pass -----------------------------------------------
// pet/rcnn/modeling/rpn/loss.py:60-141 after matching and sampling: BoxCoder.encode of the matched gt against the
// anchor (box_coder.py:13-24), smooth-L1 over the sampled positives, BCE-with-logits over all sampled anchors, both
// divided by the number of sampled anchors.  The tensor-op formulation is ~60 elementwise / reduction launches over
// 2 x 268 569 anchors forward and ~40 backward; here one thread per anchor evaluates both terms and their derivatives,
// workgroups reduce, two atomics per workgroup.  The sums are left UNDIVIDED (the sample count lives on the device).
__global__ void __launch_bounds__(256) rpn_loss_kernel(const float* __restrict__ logits, const float4* __restrict__ reg,
                                                       const float4* __restrict__ anchors,
                                                       const int64_t* __restrict__ matched,
                                                       const float4* __restrict__ gts, const int* __restrict__ gt_off,
                                                       const bool* __restrict__ pos, const bool* __restrict__ neg,
                                                       int64_t total, int per_image, float wx, float wy, float ww,
                                                       float wh, float beta, float* __restrict__ sums,
                                                       float* __restrict__ dlogits, float4* __restrict__ dreg,
                                                       const int* __restrict__ quota, int n_quota) {
  // quota (or null): the sampler's per-image (positives, negatives) counts -- both sums and both gradients leave the
  // kernel divided by their total, the loss's normalisation (rpn/loss.py:121-126), instead of by framework ops
  float inv = 1.f;
  if (quota) {
    int n = 0;
    for (int i = 0; i < n_quota; ++i) n += quota[i];
    inv = 1.f / (float)n;
  }
  float obj = 0.f, box = 0.f;
  // one anchor: its two loss terms into obj / box, its two derivatives out
  auto anchor = [&](int64_t t, bool p, bool n, float& dl, float4& dr) {
    dl = 0.f;
    dr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p || n) {
      const float x = logits[t], z = p ? 1.f : 0.f;
      obj += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));                   // binary_cross_entropy_with_logits
      dl = (1.f / (1.f + expf(-x)) - z) * inv;
    }
    if (p) {
      const float4 a = anchors[t];
      const int64_t m = matched[t] < 0 ? 0 : matched[t];
      const float4 g = gts[gt_off[(int)(t / per_image)] + m];
      const float pw = a.z - a.x + 1.f, ph = a.w - a.y + 1.f;
      const float pcx = a.x + 0.5f * pw, pcy = a.y + 0.5f * ph;
      const float gw = g.z - g.x + 1.f, gh = g.w - g.y + 1.f;
      const float gcx = g.x + 0.5f * gw, gcy = g.y + 0.5f * gh;
      const float tg[4] = {wx * (gcx - pcx) / pw, wy * (gcy - pcy) / ph, ww * logf(gw / pw), wh * logf(gh / ph)};
      const float4 r = reg[t];
      const float rv[4] = {r.x, r.y, r.z, r.w};
      float d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = rv[k] - tg[k], ae = fabsf(e);
        if (beta < 1e-5f) { box += ae; d[k] = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f); }
        else if (ae < beta) { box += 0.5f * ae * ae / beta; d[k] = e / beta; }
        else { box += ae - 0.5f * beta; d[k] = e > 0.f ? 1.f : -1.f; }
      }
      dr = make_float4(d[0] * inv, d[1] * inv, d[2] * inv, d[3] * inv);
    }
  };
  // Nearly every anchor is outside the sample (256 per image of 268 569): the kernel is a 1-byte-per-anchor read of the
  // two masks and a zero fill of the gradients.  Four anchors per thread: one 4-byte load per mask, a 16-byte store of
  // the logit gradients (the masks are torch.bool: one byte each; total's remainder is handled one by one below).
  const int64_t gtid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gstride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = ((((uintptr_t)pos | (uintptr_t)neg) & 3) == 0) && (((uintptr_t)dlogits & 15) == 0);
  const int64_t n4 = vec ? total >> 2 : 0;
  for (int64_t q = gtid; q < n4; q += gstride) {
    const uchar4 pp = ((const uchar4*)pos)[q], nn = ((const uchar4*)neg)[q];
    const bool p4[4] = {pp.x != 0, pp.y != 0, pp.z != 0, pp.w != 0}, m4[4] = {nn.x != 0, nn.y != 0, nn.z != 0, nn.w != 0};
    float dl[4];
    float4 dr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) anchor(4 * q + k, p4[k], m4[k], dl[k], dr[k]);
    ((float4*)dlogits)[q] = make_float4(dl[0], dl[1], dl[2], dl[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) dreg[4 * q + k] = dr[k];
  }
  for (int64_t t = 4 * n4 + gtid; t < total; t += gstride) {
    float dl;
    float4 dr;
    anchor(t, pos[t], neg[t], dl, dr);
    dlogits[t] = dl;
    dreg[t] = dr;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { obj += __shfl_xor(obj, d, 64); box += __shfl_xor(box, d, 64); }
  __shared__ float s_o[4], s_b[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_o[wave] = obj; s_b[wave] = box; }
  __syncthreads();
  if (threadIdx.x == 0) {
    // (most workgroups hold no sampled anchor: they add nothing, and an atomic on these two words costs every other one
    // its turn)
    const float so = s_o[0] + s_o[1] + s_o[2] + s_o[3], sb = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    if (so != 0.f) atomicAdd(&sums[0], so * inv);
    if (sb != 0.f) atomicAdd(&sums[1], sb * inv);
  }
}

}  // namespace

CPM_EXPORT int cpm_match_rois(const float* rois, const int* roi_img, const float* gts, const int* gt_off, int R,
                              int num_gts, float high, float low, int allow_low_quality, int* row_max_ws,
                              int64_t* matched, float* max_iou, void* stream) {
  CPM_REQUIRE(R >= 0 && num_gts >= 0, "negative size");
  if (R == 0) return CPM_OK;
  CPM_REQUIRE(rois && gts && gt_off && matched, "null pointer");
  CPM_REQUIRE((((uintptr_t)rois | (uintptr_t)gts) & 15) == 0, "boxes must be 16-byte aligned");
  CPM_REQUIRE(!allow_low_quality || row_max_ws, "low-quality matching needs a [num_gts] int workspace");
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((R + 255) / 256);
  if (allow_low_quality) {
    if (hipMemsetAsync(row_max_ws, 0, sizeof(int) * (size_t)(num_gts > 0 ? num_gts : 1), s) != hipSuccess)
      return CPM_ELAUNCH;
    hipLaunchKernelGGL(match_rowmax_kernel, dim3((unsigned)((R + RM_THREADS * RM_ITEMS - 1) / (RM_THREADS * RM_ITEMS))),
                       dim3(RM_THREADS), 0, s, (const float4*)rois, roi_img,
                       (const float4*)gts, gt_off, R, row_max_ws);
  }
  hipLaunchKernelGGL(match_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)rois, roi_img, (const float4*)gts,
                     gt_off, R, high, low, allow_low_quality ? row_max_ws : nullptr, matched, max_iou);
  return cpm::check_launch("match_rois");
}

CPM_EXPORT int cpm_grid_bce_loss(const float* logits, const int64_t* strides, const float* rois, const float* gt_boxes,
                                 int R, int points, int map_size, const int* sub_xy, float mapping_ratio, int radius,
                                 float weight, float* loss_sum, float* grad, void* stream) {
  CPM_REQUIRE(R >= 0, "negative size");
  if (R == 0) return CPM_OK;
  CPM_REQUIRE(logits && strides && rois && gt_boxes && sub_xy && loss_sum && grad, "null pointer");
  CPM_REQUIRE((((uintptr_t)rois | (uintptr_t)gt_boxes) & 15) == 0, "boxes must be 16-byte aligned");
  GridGeom G;
  CPM_REQUIRE(fill_geom(G, points, map_size, sub_xy, strides) == 0, "bad grid geometry");
  const float scale = weight / ((float)R * (float)points * (float)(G.half * G.half));
  hipLaunchKernelGGL(grid_bce_kernel, dim3((unsigned)(R * points)), dim3(64), 0, (hipStream_t)stream, logits,
                     (const float4*)rois, (const float4*)gt_boxes, G, mapping_ratio, radius, scale, loss_sum, grad);
  return cpm::check_launch("grid_bce_loss");
}

CPM_EXPORT int cpm_grid_decode(const float* logits, const int64_t* strides, const float* rois, int R, int points,
                               int map_size, const int* sub_xy, float mapping_ratio, const int* roi_img,
                               const float* gts, const int* gt_off, float* out_boxes, unsigned char* keep,
                               void* stream) {
  CPM_REQUIRE(R >= 0, "negative size");
  if (R == 0) return CPM_OK;
  CPM_REQUIRE(logits && strides && rois && sub_xy && out_boxes, "null pointer");
  CPM_REQUIRE(!keep || (gts && gt_off), "the gt filter needs gts and their per-image offsets");
  CPM_REQUIRE((((uintptr_t)rois | (uintptr_t)out_boxes | (uintptr_t)gts) & 15) == 0, "boxes must be 16-byte aligned");
  GridGeom G;
  CPM_REQUIRE(fill_geom(G, points, map_size, sub_xy, strides) == 0, "bad grid geometry");
  hipLaunchKernelGGL(grid_decode_kernel, dim3((unsigned)R), dim3(64 * points), 0, (hipStream_t)stream, logits,
                     (const float4*)rois, G, mapping_ratio, roi_img, (const float4*)gts, gt_off, (float4*)out_boxes,
                     keep);
  return cpm::check_launch("grid_decode");
}

CPM_EXPORT int cpm_rpn_loss(const float* logits, const float* reg, const float* anchors, const int64_t* matched,
                            const float* gts, const int* gt_off, const uint8_t* pos, const uint8_t* neg, int64_t total,
                            int per_image, const float* weights4, float beta, float* sums2, float* dlogits, float* dreg,
                            const int32_t* quota, int n_quota, void* stream) {
  CPM_REQUIRE(total >= 0 && per_image > 0 && total % per_image == 0, "bad sizes");
  CPM_REQUIRE(sums2 && weights4, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums2, 0, 2 * sizeof(float), s) != hipSuccess) return CPM_ELAUNCH;
  if (total == 0) return CPM_OK;
  CPM_REQUIRE(logits && reg && anchors && matched && gts && gt_off && pos && neg && dlogits && dreg, "null pointer");
  CPM_REQUIRE((((uintptr_t)reg | (uintptr_t)anchors | (uintptr_t)gts | (uintptr_t)dreg) & 15) == 0,
              "box tensors must be 16-byte aligned");
  int64_t b = (total / 4 + 255) / 256;                     // four anchors per thread
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(rpn_loss_kernel, dim3((unsigned)b), dim3(256), 0, s, logits, (const float4*)reg,
                     (const float4*)anchors, matched, (const float4*)gts, gt_off, (const bool*)pos, (const bool*)neg,
                     total, per_image, weights4[0], weights4[1], weights4[2], weights4[3], beta, sums2, dlogits,
                     (float4*)dreg, quota, quota ? n_quota : 0);
  return cpm::check_launch("rpn_loss");
}

CPM_EXPORT int cpm_rpn_decode(const float* reg, const int64_t* topk_idx, const float* anchors, int N, int A, int k,
                              const float* weights4, float clip, const float* im_w, const float* im_h,
                              float* out_boxes, void* stream) {
  CPM_REQUIRE(N >= 0 && A >= 0 && k >= 0 && N <= 64, "bad sizes (at most 64 images per call)");
  if (N == 0 || k == 0) return CPM_OK;
  CPM_REQUIRE(reg && topk_idx && anchors && weights4 && im_w && im_h && out_boxes, "null pointer");
  CPM_REQUIRE((((uintptr_t)reg | (uintptr_t)anchors | (uintptr_t)out_boxes) & 15) == 0, "boxes must be 16-byte aligned");
  ImSizes sz;
  for (int n = 0; n < N; ++n) { sz.w[n] = im_w[n]; sz.h[n] = im_h[n]; }
  const int total = N * k;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)reg, topk_idx, (const float4*)anchors, N, A, k, weights4[0], weights4[1],
                     weights4[2], weights4[3], clip, sz, (float4*)out_boxes);
  return cpm::check_launch("rpn_decode");
}

CPM_EXPORT int cpm_rpn_decode_multi(const float* const* reg, const int64_t* const* topk_idx,
                                    const float* const* anchors, const int* A, const int* k, const int* out_off,
                                    int levels, int N, const float* weights4, float clip, const float* im_w,
                                    const float* im_h, float* out_boxes, void* stream) {
  CPM_REQUIRE(levels >= 1 && levels <= DEC_LEVELS && N >= 1 && N <= 64, "1..8 levels, 1..64 images");
  CPM_REQUIRE(reg && topk_idx && anchors && A && k && out_off && weights4 && im_w && im_h && out_boxes, "null pointer");
  DecodeLevels lv = {};
  int kmax = 0;
  for (int l = 0; l < levels; ++l) {
    CPM_REQUIRE(reg[l] && topk_idx[l] && anchors[l] && A[l] >= 1 && k[l] >= 1 && out_off[l] >= 0, "bad level");
    CPM_REQUIRE((((uintptr_t)reg[l] | (uintptr_t)anchors[l]) & 15) == 0, "boxes must be 16-byte aligned");
    lv.reg[l] = (const float4*)reg[l]; lv.idx[l] = topk_idx[l]; lv.anchors[l] = (const float4*)anchors[l];
    lv.A[l] = A[l]; lv.k[l] = k[l]; lv.out_off[l] = out_off[l];
    if (k[l] > kmax) kmax = k[l];
  }
  CPM_REQUIRE(((uintptr_t)out_boxes & 15) == 0, "boxes must be 16-byte aligned");
  ImSizes sz;
  for (int n = 0; n < N; ++n) { sz.w[n] = im_w[n]; sz.h[n] = im_h[n]; }
  hipLaunchKernelGGL(rpn_decode_multi_kernel, dim3((unsigned)((N * kmax + 255) / 256), levels), dim3(256), 0,
                     (hipStream_t)stream, lv, N, weights4[0], weights4[1], weights4[2], weights4[3], clip, sz,
                     (float4*)out_boxes);
  return cpm::check_launch("rpn_decode_multi");
}

CPM_EXPORT int cpm_sigmoid_multi(const float* const* in, const int* n, const int* out_off, int levels, float* out,
                                 void* stream) {
  CPM_REQUIRE(levels >= 1 && levels <= DEC_LEVELS && in && n && out_off && out, "1..8 levels, non-null pointers");
  SigLevels lv = {};
  int nmax = 0;
  for (int l = 0; l < levels; ++l) {
    CPM_REQUIRE(in[l] && n[l] >= 0 && out_off[l] >= 0, "bad level");
    lv.in[l] = in[l]; lv.n[l] = n[l]; lv.out_off[l] = out_off[l];
    if (n[l] > nmax) nmax = n[l];
  }
  if (nmax == 0) return CPM_OK;
  hipLaunchKernelGGL(sigmoid_multi_kernel, dim3((unsigned)((nmax + 255) / 256), levels), dim3(256), 0,
                     (hipStream_t)stream, lv, out);
  return cpm::check_launch("sigmoid_multi");
}

CPM_EXPORT int cpm_rpn_labels(const int64_t* matched, const uint8_t* visible, int64_t total, int discard_between,
                              float* labels, void* stream) {
  CPM_REQUIRE(total >= 0, "total >= 0");
  if (total == 0) return CPM_OK;
  CPM_REQUIRE(matched && labels, "null pointer");
  hipLaunchKernelGGL(rpn_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     matched, visible, total, discard_between, labels);
  return cpm::check_launch("rpn_labels");
}

CPM_EXPORT int cpm_l2_loss_pairs(const float* x, const float* iou, const float* target, int R, float* loss,
                                 float* grad, void* stream) {
  CPM_REQUIRE(R >= 2, "at least two rows (a positive target's column index is used as a row)");
  CPM_REQUIRE(x && (iou || target) && loss && grad, "null pointer");
  CPM_REQUIRE((((uintptr_t)x | (uintptr_t)grad | (uintptr_t)target) & 7) == 0, "rows must be 8-byte aligned");
  hipLaunchKernelGGL(l2_pairs_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float2*)x, iou,
                     (const float2*)target, R, loss, (float2*)grad);
  return cpm::check_launch("l2_loss_pairs");
}

CPM_EXPORT int cpm_softmax_ce(const float* logits, const int64_t* labels, int R, int C, int64_t ignore_index,
                              float* loss, float* grad, float* row_loss, int* ticket, void* stream) {
  CPM_REQUIRE(R >= 1 && C >= 1, "at least one row and one class");
  CPM_REQUIRE(logits && labels && loss && grad && row_loss && ticket, "null pointer");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     R, C, ignore_index, loss, grad, row_loss, ticket);
  return cpm::check_launch("softmax_ce");
}
