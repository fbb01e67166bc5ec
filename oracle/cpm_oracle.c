/*
 * cpm_oracle.c -- CPU restatement of the reference's custom-op arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (cpm-r-cnn_amd/) may
 * import, link or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  Plain C99, scalar, single-threaded, fp32 with
 * contraction disabled (build with -ffp-contract=off) so that the operation
 * order -- and therefore the rounding -- is the reference's.
 *
 * Pinning:
 *   orc_roi_align_*      pinned against oracle/_ref (the reference's own
 *                        ROIAlign_cpu.cpp compiled here) + tests/golden/ops.npz
 *   orc_nms / orc_ml_nms restated from ml_nms.cu (the reference has no CPU kernel
 *                        for ml_nms, ml_nms.h:38, and nms lives in torchvision);
 *                        pinned to the keep lists of the reference's CPU greedy
 *                        NMS = soft_nms.cpp with the 'hard' method (oracle/_ref,
 *                        tie-free cases of tests/golden/soft_nms.npz) and checked
 *                        against a brute-force greedy in tests.
 *   orc_soft_nms,        pinned against oracle/_ref (the reference's own
 *   orc_ml_soft_nms      NMS/soft_nms.cpp and NMS/ml_soft_nms.cpp compiled here)
 *                        through tests/golden/soft_nms.npz
 *   orc_deform_conv      PARITY UNPINNED (CUDA-only op in the reference; see its
 *                        section header for what pins it instead).
 *   the box/level/grid helpers are pinned by goldens produced by importing the
 *   reference's Python (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* RoIAlign forward -- pet/lib/ops/csrc/ROIAlign/ROIAlign_cpu.cpp:73-166     */
/* (tap table) and :169-294 (pooling loop).  Layout NCHW, rois [K,5].         */
/* ------------------------------------------------------------------------- */
typedef struct { int p[4]; float w[4]; int valid; } orc_tap;

static void orc_bilinear_tap(int H, int W, float y, float x, orc_tap* t) {
  /* ROIAlign_cpu.cpp:101-160 */
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    t->p[0] = t->p[1] = t->p[2] = t->p[3] = 0;
    t->w[0] = t->w[1] = t->w[2] = t->w[3] = 0.f;
    t->valid = 0;
    return;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y0 = (int)y, x0 = (int)x, y1, x1;
  if (y0 >= H - 1) { y1 = y0 = H - 1; y = (float)y0; } else { y1 = y0 + 1; }
  if (x0 >= W - 1) { x1 = x0 = W - 1; x = (float)x0; } else { x1 = x0 + 1; }
  float ly = y - (float)y0, lx = x - (float)x0;
  float hy = 1.f - ly, hx = 1.f - lx;
  t->p[0] = y0 * W + x0; t->p[1] = y0 * W + x1;
  t->p[2] = y1 * W + x0; t->p[3] = y1 * W + x1;
  t->w[0] = hy * hx; t->w[1] = hy * lx; t->w[2] = ly * hx; t->w[3] = ly * lx;
  t->valid = 1;
}

static void orc_nearest_tap(int H, int W, float y, float x, orc_tap* t) {
  /* ROIAlign_cpu.cpp:47-66 and :309-318 */
  if (y < -0.5f || y >= (float)H - 0.5f || x < -0.5f || x >= (float)W - 0.5f) {
    t->p[0] = -1; t->valid = 0; return;
  }
  int xl = (int)roundf(x), yl = (int)roundf(y);
  t->p[0] = yl * W + xl; t->valid = 1;
}

typedef struct {
  float start_w, start_h, bin_w, bin_h;
  int grid_h, grid_w, batch;
} orc_roi_geom;

static void orc_roi_geometry(const float* roi, float scale, int PH, int PW,
                             int sampling_ratio, int aligned, orc_roi_geom* g) {
  /* ROIAlign_cpu.cpp:191-222 (fwd) == :407-443 (bwd) */
  g->batch = (int)roi[0];
  float off = aligned ? 0.5f : 0.0f;
  g->start_w = roi[1] * scale - off;
  g->start_h = roi[2] * scale - off;
  float end_w = roi[3] * scale - off;
  float end_h = roi[4] * scale - off;
  float rw = end_w - g->start_w, rh = end_h - g->start_h;
  if (!aligned) { rw = rw > 1.f ? rw : 1.f; rh = rh > 1.f ? rh : 1.f; }
  g->bin_h = rh / (float)PH;
  g->bin_w = rw / (float)PW;
  g->grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  g->grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
}

ORC_API int orc_roi_align_forward(const float* input, const float* rois, int K,
                                  int B, int C, int H, int W, float scale,
                                  int PH, int PW, int sampling_ratio,
                                  int aligned, int interp, float* out) {
  if (interp != 0 && interp != 1) return -1;
  for (int n = 0; n < K; ++n) {
    orc_roi_geom g;
    orc_roi_geometry(rois + 5 * n, scale, PH, PW, sampling_ratio, aligned, &g);
    if (g.batch < 0 || g.batch >= B) return -2;
    if (aligned && (g.bin_w < 0.f || g.bin_h < 0.f)) return -3;
    int ng = g.grid_h * g.grid_w;
    /* ROIAlign_cpu.cpp:219: count = max(grid_h*grid_w, 1) */
    float count = (float)(ng > 1 ? ng : 1);
    orc_tap* taps = (orc_tap*)malloc(sizeof(orc_tap) * (size_t)(ng > 0 ? ng : 1) * PH * PW);
    int ti = 0;
    for (int ph = 0; ph < PH; ++ph)
      for (int pw = 0; pw < PW; ++pw)
        for (int iy = 0; iy < g.grid_h; ++iy) {
          /* ROIAlign_cpu.cpp:91-93: same expression tree */
          float yy = g.start_h + (float)ph * g.bin_h +
                     ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
          for (int ix = 0; ix < g.grid_w; ++ix) {
            float xx = g.start_w + (float)pw * g.bin_w +
                       ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
            if (interp == 0) orc_bilinear_tap(H, W, yy, xx, &taps[ti++]);
            else orc_nearest_tap(H, W, yy, xx, &taps[ti++]);
          }
        }
    for (int c = 0; c < C; ++c) {
      const float* src = input + ((size_t)g.batch * C + c) * H * W;
      float* dst = out + (((size_t)n * C + c) * PH) * PW;
      ti = 0;
      for (int b = 0; b < PH * PW; ++b) {
        float acc = 0.f;
        for (int s = 0; s < ng; ++s, ++ti) {
          const orc_tap* t = &taps[ti];
          if (interp == 0) {
            /* ROIAlign_cpu.cpp:271-273: ((w1 v1 + w2 v2) + w3 v3) + w4 v4 */
            acc += t->w[0] * src[t->p[0]] + t->w[1] * src[t->p[1]] +
                   t->w[2] * src[t->p[2]] + t->w[3] * src[t->p[3]];
          } else if (t->p[0] >= 0) {
            acc += src[t->p[0]];
          }
        }
        dst[b] = acc / count;
      }
    }
    free(taps);
  }
  return 0;
}

/* RoIAlign backward -- ROIAlign_cpu.cpp:387-496.  grad [K,C,PH,PW] contiguous */
ORC_API int orc_roi_align_backward(const float* grad, const float* rois, int K,
                                   int B, int C, int H, int W, float scale,
                                   int PH, int PW, int sampling_ratio,
                                   int aligned, int interp, float* grad_input) {
  if (interp != 0 && interp != 1) return -1;
  memset(grad_input, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int n = 0; n < K; ++n) {
    orc_roi_geom g;
    orc_roi_geometry(rois + 5 * n, scale, PH, PW, sampling_ratio, aligned, &g);
    if (g.batch < 0 || g.batch >= B) return -2;
    float count = (float)(g.grid_h * g.grid_w);   /* :446 (no max here) */
    for (int c = 0; c < C; ++c) {
      float* dst = grad_input + ((size_t)g.batch * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float go = grad[(((size_t)n * C + c) * PH + ph) * PW + pw];
          for (int iy = 0; iy < g.grid_h; ++iy) {
            float y = g.start_h + (float)ph * g.bin_h +
                      ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
              float x = g.start_w + (float)pw * g.bin_w +
                        ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
              orc_tap t;
              if (interp == 0) {
                orc_bilinear_tap(H, W, y, x, &t);
                if (!t.valid) continue;       /* :341-346 sets indices to -1 */
                /* :465-468: g_i = go * w_i / count */
                dst[t.p[0]] += go * t.w[0] / count;
                dst[t.p[1]] += go * t.w[1] / count;
                dst[t.p[2]] += go * t.w[2] / count;
                dst[t.p[3]] += go * t.w[3] / count;
              } else {
                orc_nearest_tap(H, W, y, x, &t);
                if (t.valid) dst[t.p[0]] += go / count;
              }
            }
          }
        }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* NMS.  ml_nms: pet/lib/ops/csrc/NMS/ml_nms.cu:11-26 (IoU, same-label test), */
/* :92-94 (sort by score, descending), :127-140 (greedy sweep, topk early-out) */
/* :143-145 (indices into the caller's order).  Ties in the sort are broken   */
/* by ascending original index (a stable sort); the reference leaves this     */
/* unspecified.                                                               */
/* ------------------------------------------------------------------------- */
typedef struct { float s; int i; } orc_si;
static int orc_si_cmp(const void* a, const void* b) {
  const orc_si* x = (const orc_si*)a; const orc_si* y = (const orc_si*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

static int orc_iou_gt(const float* a, const float* b, float thr) {
  /* ml_nms.cu:19-25 -- areas without the "+1" */
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  float inter = w * h;
  float sa = (a[2] - a[0]) * (a[3] - a[1]);
  float sb = (b[2] - b[0]) * (b[3] - b[1]);
  return (inter / (sa + sb - inter)) > thr;
}

ORC_API int64_t orc_ml_nms(const float* boxes, const float* scores,
                           const int64_t* labels, int64_t n, float thr,
                           int64_t topk, int64_t* keep) {
  if (n <= 0) return 0;
  orc_si* ord = (orc_si*)malloc(sizeof(orc_si) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) { ord[i].s = scores[i]; ord[i].i = (int)i; }
  qsort(ord, (size_t)n, sizeof(orc_si), orc_si_cmp);
  unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
  int64_t nk = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep[nk++] = ord[i].i;
    if (nk == topk) break;                       /* ml_nms.cu:134 */
    const float* a = boxes + 4 * (size_t)ord[i].i;
    for (int64_t j = i + 1; j < n; ++j) {
      if (dead[j]) continue;
      if (labels && labels[ord[i].i] != labels[ord[j].i]) continue;   /* :16 */
      if (orc_iou_gt(a, boxes + 4 * (size_t)ord[j].i, thr)) dead[j] = 1;
    }
  }
  free(dead); free(ord);
  return nk;
}

/* Single-class NMS: pet/lib/ops/nms.py:2,10 binds torchvision.ops.nms.        */
/* Published semantics: sort by score descending, greedy, suppress IoU > thr,  */
/* IoU without +1, return kept indices in score order.                         */
ORC_API int64_t orc_nms(const float* boxes, const float* scores, int64_t n,
                        float thr, int64_t* keep) {
  return orc_ml_nms(boxes, scores, NULL, n, thr, 0, keep);
}

/* Soft-NMS -- pet/lib/ops/csrc/NMS/soft_nms.cpp:5-110 and its multi-label twin        */
/* NMS/ml_soft_nms.cpp:5-122 (only boxes of the selected box's label decay; stop after   */
/* `topk` selections -- `if (topk == i)`, so topk 0 returns nothing and topk < 0 never   */
/* stops).  boxes [n,4], scores [n], labels [n] (NULL: single label) are reordered IN    */
/* PLACE as the reference does; idx[n] receives the original indices; returns the number */
/* of survivors (the first `ret` rows).  method 1 linear, 2 gaussian, otherwise hard.     */
ORC_API int64_t orc_ml_soft_nms(float* boxes, float* scores, int64_t* labels, int64_t* idx, int64_t n, float thr,
                                int method, float sigma, float min_score, int64_t topk) {
  float* area = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) {
    area[i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);   /* :22 */
    idx[i] = i;
  }
  int64_t nd = n;
  for (int64_t i = 0; i < nd; ++i) {
    if (labels && topk == i) { nd = topk; break; }              /* ml_soft_nms.cpp:31-35 */
    int64_t mp = i;                                             /* :30-38 first position of the maximum */
    float ms = scores[i];
    for (int64_t q = i + 1; q < nd; ++q)
      if (ms < scores[q]) { ms = scores[q]; mp = q; }
    float b[4], sc = scores[mp], ar = area[mp];                 /* :41-66 swap to the front */
    int64_t id = idx[mp], lab = labels ? labels[mp] : 0;
    memcpy(b, boxes + 4 * mp, sizeof b);
    memcpy(boxes + 4 * mp, boxes + 4 * i, sizeof b);
    scores[mp] = scores[i]; area[mp] = area[i]; idx[mp] = idx[i];
    if (labels) { labels[mp] = labels[i]; labels[i] = lab; }
    memcpy(boxes + 4 * i, b, sizeof b);
    scores[i] = sc; area[i] = ar; idx[i] = id;
    for (int64_t q = i + 1; q < nd; ++q) {                      /* :70-108 decay, drop by swapping with the last */
      if (!labels || labels[q] == lab) {
        const float* c = boxes + 4 * q;
        float inter = fmaxf(0.f, fminf(b[2], c[2]) - fmaxf(b[0], c[0])) *
                      fmaxf(0.f, fminf(b[3], c[3]) - fmaxf(b[1], c[1]));
        float ovr = inter / (ar + area[q] - inter);
        if (method == 1) {
          if (ovr > thr) scores[q] = (1.f - ovr) * scores[q];
        } else if (method == 2) {
          scores[q] = expf(-(ovr * ovr) / sigma) * scores[q];
        } else {
          if (ovr > thr) scores[q] = 0.f;
        }
      }
      if (scores[q] < min_score) {
        --nd;
        memcpy(boxes + 4 * q, boxes + 4 * nd, sizeof b);
        scores[q] = scores[nd]; area[q] = area[nd]; idx[q] = idx[nd];
        if (labels) labels[q] = labels[nd];
        --q;
      }
    }
  }
  free(area);
  return nd;
}

ORC_API int64_t orc_soft_nms(float* boxes, float* scores, int64_t* idx, int64_t n, float thr, int method,
                             float sigma, float min_score) {
  return orc_ml_soft_nms(boxes, scores, NULL, idx, n, thr, method, sigma, min_score, -1);
}

/* box_iou -- pet/lib/ops/csrc/Box_ops/box_iou.cu:27-77 (no +1), out [N,K]     */
ORC_API void orc_box_iou(const float* boxes, int64_t N, const float* query,
                         int64_t K, float* out) {
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = 0; j < K; ++j) {
      const float* a = boxes + 4 * i; const float* b = query + 4 * j;
      float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
      float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
      float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
      float inter = w * h;
      float sa = (a[2] - a[0]) * (a[3] - a[1]);
      float sb = (b[2] - b[0]) * (b[3] - b[1]);
      out[i * K + j] = inter / (sa + sb - inter);
    }
}

/* boxlist_iou -- pet/utils/data/structures/boxlist_ops.py:123-158 ("+1")      */
/* and BoxList.area bounding_box.py:306-316.  out [N,M]                        */
ORC_API void orc_boxlist_iou(const float* b1, int64_t N, const float* b2,
                             int64_t M, float* out) {
  for (int64_t i = 0; i < N; ++i) {
    const float* a = b1 + 4 * i;
    float area1 = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    for (int64_t j = 0; j < M; ++j) {
      const float* b = b2 + 4 * j;
      float area2 = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
      float ltx = fmaxf(a[0], b[0]), lty = fmaxf(a[1], b[1]);
      float rbx = fminf(a[2], b[2]), rby = fminf(a[3], b[3]);
      float w = rbx - ltx + 1.f; if (w < 0.f) w = 0.f;
      float h = rby - lty + 1.f; if (h < 0.f) h = 0.f;
      float inter = w * h;
      out[i * M + j] = inter / (area1 + area2 - inter);
    }
  }
}

/* PoolPointsInterp -- pet/lib/ops/csrc/PoolPointsInterp/PoolPointsInterp_cuda.cu */
/* :12-60 (bilinear), :62-91 (fwd; batch index = n / 196, :74), :147-196 (bwd)  */
ORC_API void orc_pool_points_interp_forward(const float* input, const float* pts,
                                            int K, int C, int H, int W,
                                            float scale, float* out) {
  for (int n = 0; n < K; ++n) {
    int b = n / 196;
    float X = pts[3 * n + 1] * scale, Y = pts[3 * n + 2] * scale;
    orc_tap t; orc_bilinear_tap(H, W, Y, X, &t);
    for (int c = 0; c < C; ++c) {
      const float* src = input + ((size_t)b * C + c) * H * W;
      float v = 0.f;
      if (t.valid)
        v = t.w[0] * src[t.p[0]] + t.w[1] * src[t.p[1]] +
            t.w[2] * src[t.p[2]] + t.w[3] * src[t.p[3]];
      out[(size_t)n * C + c] = v;
    }
  }
}

ORC_API void orc_pool_points_interp_backward(const float* grad, const float* pts,
                                             int K, int B, int C, int H, int W,
                                             float scale, float* gin) {
  memset(gin, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int n = 0; n < K; ++n) {
    int b = n / 196;
    float X = pts[3 * n + 1] * scale, Y = pts[3 * n + 2] * scale;
    orc_tap t; orc_bilinear_tap(H, W, Y, X, &t);
    if (!t.valid) continue;
    for (int c = 0; c < C; ++c) {
      float* dst = gin + ((size_t)b * C + c) * H * W;
      float go = grad[(size_t)n * C + c];
      for (int q = 0; q < 4; ++q) dst[t.p[q]] += go * t.w[q];
    }
  }
}

/* LevelMapper -- pet/rcnn/utils/poolers.py:30-40 with BoxList.area (+1)       */
/* lvl = clamp(floor(lvl0 + log2(sqrt(area)/s0 + eps)), kmin, kmax) - kmin     */
ORC_API void orc_level_map(const float* boxes, int64_t n, float k_min,
                           float k_max, float s0, float lvl0, float eps,
                           int64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float* b = boxes + 4 * i;
    float area = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    float s = sqrtf(area);
    float l = floorf(lvl0 + log2f(s / s0 + eps));
    if (l < k_min) l = k_min;
    if (l > k_max) l = k_max;
    out[i] = (int64_t)l - (int64_t)k_min;
  }
}

/* BoxCoder.decode -- pet/rcnn/utils/box_coder.py:51-94 (single 4-column code) */
ORC_API void orc_box_decode(const float* codes, const float* boxes, int64_t n,
                            float wx, float wy, float ww, float wh, float clip,
                            float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float* b = boxes + 4 * i; const float* c = codes + 4 * i;
    float w = b[2] - b[0] + 1.f, h = b[3] - b[1] + 1.f;
    float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
    float dx = c[0] / wx, dy = c[1] / wy, dw = c[2] / ww, dh = c[3] / wh;
    if (dw > clip) dw = clip;
    if (dh > clip) dh = clip;
    float pcx = dx * w + cx, pcy = dy * h + cy;
    float pw = expf(dw) * w, ph = expf(dh) * h;
    out[4 * i + 0] = pcx - 0.5f * pw;
    out[4 * i + 1] = pcy - 0.5f * ph;
    out[4 * i + 2] = pcx + 0.5f * pw - 1.f;
    out[4 * i + 3] = pcy + 0.5f * ph - 1.f;
  }
}

/* BoxCoder.encode -- box_coder.py:21-49 */
ORC_API void orc_box_encode(const float* ref, const float* prop, int64_t n,
                            float wx, float wy, float ww, float wh, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float* p = prop + 4 * i; const float* g = ref + 4 * i;
    float ew = p[2] - p[0] + 1.f, eh = p[3] - p[1] + 1.f;
    float ecx = p[0] + 0.5f * ew, ecy = p[1] + 0.5f * eh;
    float gw = g[2] - g[0] + 1.f, gh = g[3] - g[1] + 1.f;
    float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
    out[4 * i + 0] = wx * (gcx - ecx) / ew;
    out[4 * i + 1] = wy * (gcy - ecy) / eh;
    out[4 * i + 2] = ww * logf(gw / ew);
    out[4 * i + 3] = wh * logf(gh / eh);
  }
}

/* Matcher -- pet/rcnn/utils/matcher.py:41-111.  q [M,N] (gt x predictions)   */
ORC_API void orc_matcher(const float* q, int64_t M, int64_t N, float high,
                         float low, int allow_low_quality, int64_t* matches) {
  int64_t* all = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
  for (int64_t j = 0; j < N; ++j) {
    float best = q[j]; int64_t bi = 0;
    for (int64_t i = 1; i < M; ++i)
      if (q[i * N + j] > best) { best = q[i * N + j]; bi = i; }   /* first max */
    all[j] = bi;
    if (best < low) matches[j] = -1;                 /* BELOW_LOW_THRESHOLD */
    else if (best < high) matches[j] = -2;           /* BETWEEN_THRESHOLDS  */
    else matches[j] = bi;
  }
  if (allow_low_quality) {
    /* matcher.py:88-111: for every gt, every prediction tying its row max */
    for (int64_t i = 0; i < M; ++i) {
      float mx = q[i * N];
      for (int64_t j = 1; j < N; ++j) if (q[i * N + j] > mx) mx = q[i * N + j];
      for (int64_t j = 0; j < N; ++j) if (q[i * N + j] == mx) matches[j] = all[j];
    }
  }
  free(all);
}

/* calc_sub_regions -- pet/rcnn/modeling/grid_rcnn/loss.py:244-273 /           */
/* grid_cascade_rcnn/loss.py:279-308.  out [grid_points,4] = x1,y1,x2,y2       */
ORC_API void orc_sub_regions(int grid_points, int grid_size, int map_size,
                             int* out) {
  int half = map_size / 4 * 2;
  for (int i = 0; i < grid_points; ++i) {
    int xi = i / grid_size, yi = i % grid_size, sx, sy;
    if (xi == 0) sx = 0;
    else if (xi == grid_size - 1) sx = half;
    else { double r = (double)xi / (grid_size - 1) - 0.25; sx = (int)(r * map_size); if (sx < 0) sx = 0; }
    if (yi == 0) sy = 0;
    else if (yi == grid_size - 1) sy = half;
    else { double r = (double)yi / (grid_size - 1) - 0.25; sy = (int)(r * map_size); if (sy < 0) sy = 0; }
    out[4 * i + 0] = sx; out[4 * i + 1] = sy;
    out[4 * i + 2] = sx + half; out[4 * i + 3] = sy + half;
  }
}

/* Grid heat-map targets -- grid_cascade_rcnn/loss.py:178-258.                 */
/* boxes/gt [R,4]; out [R, P, half, half] (already cropped to the sub-regions) */
ORC_API void orc_grid_targets(const float* boxes, const float* gt, int64_t R,
                              int grid_points, int map_size, int radius,
                              float mapping_ratio, float* out) {
  int gs = (int)(sqrt((double)grid_points));
  int half = map_size / 4 * 2;
  int* sub = (int*)malloc(sizeof(int) * 4 * grid_points);
  orc_sub_regions(grid_points, gs, map_size, sub);
  float* full = (float*)malloc(sizeof(float) * (size_t)map_size * map_size);
  for (int64_t i = 0; i < R; ++i) {
    const float* b = boxes + 4 * i; const float* g = gt + 4 * i;
    /* loss.py:186-193: expand the RoI by mapping_ratio */
    float x1 = b[0] - mapping_ratio * ((b[2] - b[0]) / 2.f);
    float y1 = b[1] - mapping_ratio * ((b[3] - b[1]) / 2.f);
    float x2 = b[2] + mapping_ratio * ((b[2] - b[0]) / 2.f);
    float y2 = b[3] + mapping_ratio * ((b[3] - b[1]) / 2.f);
    float bw = x2 - x1, bh = y2 - y1;
    int skip = (bw <= (float)gs || bh <= (float)gs);          /* :215-217 */
    for (int j = 0; j < grid_points; ++j) {
      memset(full, 0, sizeof(float) * (size_t)map_size * map_size);
      if (!skip) {
        /* :205-209 factors; python-float factors are exactly fp32 for 2/3x3 */
        float fx = (float)(1.0 - (double)(j / gs) / (gs - 1));
        float fy = (float)(1.0 - (double)(j % gs) / (gs - 1));
        float gx = fx * g[0] + (1.f - fx) * g[2];
        float gy = fy * g[1] + (1.f - fy) * g[3];
        /* :226-229 python int(): truncation toward zero of the fp32 value */
        int cx = (int)((gx - x1) / bw * (float)map_size);
        int cy = (int)((gy - y1) / bh * (float)map_size);
        for (int x = cx - radius; x <= cx + radius; ++x)
          for (int y = cy - radius; y <= cy + radius; ++y)
            if (x >= 0 && x < map_size && y >= 0 && y < map_size &&
                (x - cx) * (x - cx) + (y - cy) * (y - cy) <= radius * radius)
              full[y * map_size + x] = 1.f;
      }
      /* :252-256 crop the point's sub-region */
      int sx = sub[4 * j], sy = sub[4 * j + 1];
      float* dst = out + (((size_t)i * grid_points + j) * half) * half;
      for (int y = 0; y < half; ++y)
        for (int x = 0; x < half; ++x)
          dst[y * half + x] = full[(sy + y) * map_size + sx + x];
    }
  }
  free(full); free(sub);
}

/* Grid -> box decoder -- grid_cascade_rcnn/inference.py:189-279 (get_boxes).  */
/* prob [R,P,h,w] = sigmoid(logits) (the sigmoid itself is torch arithmetic).  */
/* No clipping: inference.py:275-276 clamps a copy.           */
ORC_API void orc_grid_decode(const float* boxes, const float* prob, int64_t R,
                             int grid_points, int map_size,
                             float mapping_ratio, float* out) {
  int gs = (int)(sqrt((double)grid_points));
  int half = map_size / 4 * 2;
  int* sub = (int*)malloc(sizeof(int) * 4 * grid_points);
  orc_sub_regions(grid_points, gs, map_size, sub);
  float* sc = (float*)malloc(sizeof(float) * grid_points);
  float* ax = (float*)malloc(sizeof(float) * grid_points);
  float* ay = (float*)malloc(sizeof(float) * grid_points);
  for (int64_t i = 0; i < R; ++i) {
    const float* b = boxes + 4 * i;
    float wdt = b[2] - b[0], hgt = b[3] - b[1];
    float x1 = b[0] - mapping_ratio * (wdt / 2.f);
    float y1 = b[1] - mapping_ratio * (hgt / 2.f);
    for (int j = 0; j < grid_points; ++j) {
      const float* p = prob + (((size_t)i * grid_points + j) * half) * half;
      float best = p[0]; int bi = 0;
      for (int k = 1; k < half * half; ++k) if (p[k] > best) { best = p[k]; bi = k; }
      int xs = bi % half + sub[4 * j], ys = bi / half + sub[4 * j + 1];
      sc[j] = best;
      /* :239-248 -- ((xs+.5)/(2w)) * (1+ratio) * width + x1, left to right */
      ax[j] = ((float)xs + 0.5f) / (float)(2 * half) * (1.f + mapping_ratio) * wdt + x1;
      ay[j] = ((float)ys + 0.5f) / (float)(2 * half) * (1.f + mapping_ratio) * hgt + y1;
    }
    /* :251-273 score-weighted vote of the gs points on each side */
    float nx1 = 0, dx1 = 0, ny1 = 0, dy1 = 0, nx2 = 0, dx2 = 0, ny2 = 0, dy2 = 0;
    for (int k = 0; k < gs; ++k) {
      int ix1 = k, iy1 = k * gs, ix2 = grid_points - gs + k, iy2 = (k + 1) * gs - 1;
      nx1 += ax[ix1] * sc[ix1]; dx1 += sc[ix1];
      ny1 += ay[iy1] * sc[iy1]; dy1 += sc[iy1];
      nx2 += ax[ix2] * sc[ix2]; dx2 += sc[ix2];
      ny2 += ay[iy2] * sc[iy2]; dy2 += sc[iy2];
    }
    out[4 * i + 0] = nx1 / dx1; out[4 * i + 1] = ny1 / dy1;
    out[4 * i + 2] = nx2 / dx2; out[4 * i + 3] = ny2 / dy2;
  }
  free(sc); free(ax); free(ay); free(sub);
}

/* generate_anchors -- pet/rcnn/modeling/rpn/anchor_generator.py:221-290       */
/* (double precision like numpy, then cast to fp32 by the caller). out [S*A,4] */
ORC_API void orc_cell_anchors(double stride, const double* sizes, int ns,
                              const double* ratios, int nr, double* out) {
  double base[4] = {0, 0, stride - 1, stride - 1};
  double w = base[2] - base[0] + 1, h = base[3] - base[1] + 1;
  double xc = base[0] + 0.5 * (w - 1), yc = base[1] + 0.5 * (h - 1);
  int o = 0;
  for (int r = 0; r < nr; ++r) {
    double size = w * h, sr = size / ratios[r];
    double ws = nearbyint(sqrt(sr)), hs = nearbyint(ws * ratios[r]);  /* np.round = half-even */
    double a[4] = {xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)};
    double aw = a[2] - a[0] + 1, ah = a[3] - a[1] + 1;
    double axc = a[0] + 0.5 * (aw - 1), ayc = a[1] + 0.5 * (ah - 1);
    for (int s = 0; s < ns; ++s) {
      double sc = sizes[s] / stride;
      double sw = aw * sc, sh = ah * sc;
      out[o++] = axc - 0.5 * (sw - 1); out[o++] = ayc - 0.5 * (sh - 1);
      out[o++] = axc + 0.5 * (sw - 1); out[o++] = ayc + 0.5 * (sh - 1);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Deformable convolution v1 -- pet/lib/ops/csrc/Deformable/                 */
/*   deform_conv_cuda_kernel.cu:95-128 (bilinear sample), :215-287 (im2col), */
/*   :131-160 (gradient weight), :163-213 (coordinate weight), :290-460       */
/*   (col2im / col2im_coord); contraction deform_conv_cuda.cu:402-407.        */
/* PARITY UNPINNED against the reference binary (CUDA-only op, no CPU kernel  */
/* and no test vectors in the reference); pinned instead by a known answer    */
/* (zero offsets == grouped convolution) and an independent autograd          */
/* formulation in tests/test_deform_oracle.py.                                */
/* Layout NCHW; offset [N, dg*2*R*S, P, Q]; weight [K, C/groups, R, S].       */
/* mode 0: y only.  mode 1: also dx, doffset, dw from dy (all accumulate into */
/* zero-filled outputs).                                                      */
/* ------------------------------------------------------------------------- */
typedef struct { int ok; int h0, w0; float lh, lw; } orc_dsample;

static orc_dsample orc_deform_pos(const float* offset, int n, int dg_n, int dgi, int R, int S, int P, int Q, int i,
                                  int j, int p, int q, int stride, int pad, int dil, int H, int W, float* hh,
                                  float* ww) {
  orc_dsample s;
  const int ch = ((n * dg_n + dgi) * R * S + i * S + j) * 2;
  const float oh = offset ? offset[((size_t)ch * P + p) * Q + q] : 0.f;
  const float ow = offset ? offset[((size_t)(ch + 1) * P + p) * Q + q] : 0.f;
  const float h = (float)(p * stride - pad + i * dil) + oh;
  const float w = (float)(q * stride - pad + j * dil) + ow;
  s.ok = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
  s.h0 = (int)floorf(h);
  s.w0 = (int)floorf(w);
  s.lh = h - (float)s.h0;
  s.lw = w - (float)s.w0;
  *hh = h; *ww = w;
  return s;
}

ORC_API void orc_deform_conv(const float* x, const float* offset, const float* wt, const float* dy, int N, int C,
                             int H, int W, int K, int R, int S, int stride, int pad, int dil, int groups, int dg,
                             int mode, float* y, float* dx, float* doffset, float* dw) {
  const int P = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  const int Q = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  const int Cg = C / groups, Kg = K / groups, Cd = C / dg;
  for (int n = 0; n < N; ++n)
    for (int p = 0; p < P; ++p)
      for (int q = 0; q < Q; ++q)
        for (int c = 0; c < C; ++c) {
          const int g = c / Cg, cl = c - g * Cg, dgi = c / Cd;
          const float* im = x + ((size_t)n * C + c) * H * W;
          for (int i = 0; i < R; ++i)
            for (int j = 0; j < S; ++j) {
              float h, w;
              const orc_dsample s = orc_deform_pos(offset, n, dg, dgi, R, S, P, Q, i, j, p, q, stride, pad, dil, H,
                                                   W, &h, &w);
              float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, val = 0.f;
              const int h1 = s.h0 + 1, w1 = s.w0 + 1;
              const float hh = 1.f - s.lh, hw = 1.f - s.lw;
              if (s.ok) {
                if (s.h0 >= 0 && s.w0 >= 0) v1 = im[s.h0 * W + s.w0];
                if (s.h0 >= 0 && w1 <= W - 1) v2 = im[s.h0 * W + w1];
                if (h1 <= H - 1 && s.w0 >= 0) v3 = im[h1 * W + s.w0];
                if (h1 <= H - 1 && w1 <= W - 1) v4 = im[h1 * W + w1];
                val = hh * hw * v1 + hh * s.lw * v2 + s.lh * hw * v3 + s.lh * s.lw * v4;
              }
              double gcol = 0.0;
              for (int kl = 0; kl < Kg; ++kl) {
                const int k = g * Kg + kl;
                const size_t wi = (((size_t)k * Cg + cl) * R + i) * S + j;
                const size_t yi = (((size_t)n * K + k) * P + p) * Q + q;
                if (mode == 0) {
                  y[yi] += wt[wi] * val;
                } else {
                  y[yi] += wt[wi] * val;
                  gcol += (double)wt[wi] * dy[yi];
                  dw[wi] += dy[yi] * val;
                }
              }
              if (mode == 1 && s.ok) {
                const float gc = (float)gcol;
                float* dim = dx + ((size_t)n * C + c) * H * W;
                if (s.h0 >= 0 && s.w0 >= 0) dim[s.h0 * W + s.w0] += hh * hw * gc;
                if (s.h0 >= 0 && w1 <= W - 1) dim[s.h0 * W + w1] += hh * s.lw * gc;
                if (h1 <= H - 1 && s.w0 >= 0) dim[h1 * W + s.w0] += s.lh * hw * gc;
                if (h1 <= H - 1 && w1 <= W - 1) dim[h1 * W + w1] += s.lh * s.lw * gc;
                if (offset) {
                  const int ch = ((n * dg + dgi) * R * S + i * S + j) * 2;
                  /* get_coordinate_weight, bp_dir 0 (rows) and 1 (columns) */
                  const float ch_w = -hw * v1 - s.lw * v2 + hw * v3 + s.lw * v4;
                  const float cw_w = -hh * v1 + hh * v2 - s.lh * v3 + s.lh * v4;
                  doffset[((size_t)ch * P + p) * Q + q] += ch_w * gc;
                  doffset[((size_t)(ch + 1) * P + p) * Q + q] += cw_w * gc;
                }
              }
            }
        }
}
