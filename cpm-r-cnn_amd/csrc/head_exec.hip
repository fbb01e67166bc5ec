// Native executor for a chain of RoI-head layers, each  conv + bias -> [GroupNorm] -> [ReLU]:
//   the CMM grid head   8 x [conv3x3 -> GroupNorm -> ReLU]          (grid_rcnn/heads/grid_heads.py:41-57,146-152)
//   the cls / RSM head  fc6 -> ReLU -> fc7 -> ReLU -> cls_score      (heads/cls_heads.py:13-48, outputs.py:87-104)
//   the ISM branch      iou_fc1 -> ReLU -> iou_fc2 -> ReLU -> iou_pred (outputs.py:38-45,76-83)
// (a Linear is a 1x1 conv on a 1x1 image, fc6 / iou_fc1 a full-window conv on the 7x7 RoI map).
//
// With a few dozen RoIs per stage these layers are launch bound: every kernel runs 3-20 us while the Python path
// around it (autograd Function, tensor allocation, ctypes marshalling) costs 25 us per op forward and 60-100 us
// backward -- and the backward pass STARTS with the smallest of them.  The entry points below run a whole chain from
// ONE call per direction: the same C-ABI kernels in the same order with the same arguments as the per-op path (so the
// results are the same), the host cost of up to 16 ops replaced by a loop in C.  Host code only: no kernels of its own.
#include "common.h"
#include <cstdlib>

namespace {

// CPM_CHAIN_FILL=0: every split launch seeds / clears its own output (the A/B of the GroupNorm side jobs); read per call
inline bool chain_fill() {
  const char* v = getenv("CPM_CHAIN_FILL");
  return !(v && v[0] == '0');
}

inline int n_pix(const cpm_conv_desc& d) { return d.P * d.Q; }
// a full-window conv (an FC over a flattened NHWC map) seen as the 1x1 problem over R*S*C "channels" it is: every
// input pixel meets exactly one tap, the KRSC weight bytes already are that [K, R*S*C] matrix
inline cpm_conv_desc flat_desc(const cpm_conv_desc& d) {
  cpm_conv_desc f = {};
  f.N = d.N; f.H = 1; f.W = 1; f.C = d.C * d.H * d.W;
  f.K = d.K; f.R = 1; f.S = 1;
  f.stride = 1; f.pad = 0; f.dilation = 1; f.groups = 1;
  f.P = 1; f.Q = 1;
  return f;
}
inline size_t al(size_t v) { return (v + 63) / 64 * 64; }          // 256-byte aligned pieces

// Where layer i's tensors live inside the two caller-owned buffers for a batch of N samples (float offsets); -1 = the
// separate tensor (the chain's output y / the input gradient dx), -2 = the layer has no such tensor.
struct Off { int64_t conv_out, gn_out, mean, rstd, d_conv, d_in; };

constexpr int MAX_LAYERS = 64;

void layout(const cpm_chain_layer* layers, int n_layers, int N, Off* off, size_t* fwd_floats, size_t* bwd_floats,
            size_t* ws_bytes) {
  size_t f = 0, b = 0, ws = 0;
  for (int i = 0; i < n_layers; ++i) {
    const cpm_chain_layer& L = layers[i];
    cpm_conv_desc d = L.conv;
    d.N = N;
    const size_t out_n = (size_t)N * d.K * d.P * d.Q, in_n = (size_t)N * d.C * d.H * d.W;
    const size_t st = (size_t)N * L.gn_groups;
    const bool last = i == n_layers - 1;
    Off o = {-2, -2, -2, -2, -2, -2};
    if (L.has_gn) {
      o.conv_out = (int64_t)f; f += al(out_n);
      o.gn_out = -1;
      if (!last) { o.gn_out = (int64_t)f; f += al(out_n); }
      o.mean = (int64_t)f; f += al(st);
      o.rstd = (int64_t)f; f += al(st);
      o.d_conv = (int64_t)b; b += al(out_n);
    } else {
      o.conv_out = -1;
      if (!last) { o.conv_out = (int64_t)f; f += al(out_n); }
    }
    o.d_in = -1;
    if (i > 0) { o.d_in = (int64_t)b; b += al(in_n); }
    if (off) off[i] = o;
    if (ws_bytes) {
      size_t w = cpm_conv2d_workspace_bytes(&d);
      if (w > ws) ws = w;
      if (L.dgrad_flat) {
        const cpm_conv_desc df = flat_desc(d);
        w = cpm_conv2d_workspace_bytes(&df);
        if (w > ws) ws = w;
      }
    }
  }
  if (fwd_floats) *fwd_floats = f > 0 ? f : 64;
  if (bwd_floats) *bwd_floats = b > 0 ? b : 64;
  if (ws_bytes) *ws_bytes = ws;
}

inline float* at(float* base, int64_t off, float* dflt) { return off >= 0 ? base + off : dflt; }
// the tensor the NEXT layer reads: GroupNorm's output, or the conv's own (with its ReLU applied by the epilogue)
inline float* out_of(const cpm_chain_layer& L, const Off& o, float* base, float* y) {
  return L.has_gn ? at(base, o.gn_out, y) : at(base, o.conv_out, y);
}

int check_table(const cpm_chain_layer* layers, int n_layers) {
  for (int i = 0; i < n_layers; ++i) {
    const cpm_chain_layer& L = layers[i];
    if (!L.w) return 1;
    if (L.has_gn && !(L.gamma && L.beta && L.gn_groups >= 1)) return 2;
    // a conv-epilogue ReLU is undone by the NEXT layer's data gradient (gate on its input): the last layer has none
    if (!L.has_gn && L.relu && i == n_layers - 1) return 3;
    if (L.dgrad_flat && !(L.conv.R == L.conv.H && L.conv.S == L.conv.W && L.conv.pad == 0 && L.conv.stride == 1 &&
                          L.conv.groups == 1))
      return 4;
  }
  return 0;
}

}  // namespace

CPM_EXPORT int cpm_layer_chain_sizes(const cpm_chain_layer* layers, int n_layers, int N, size_t* fwd_floats,
                                     size_t* bwd_floats, size_t* workspace_bytes) {
  CPM_REQUIRE(layers && n_layers >= 1 && n_layers <= MAX_LAYERS && N >= 1, "1..64 layers, N >= 1");
  CPM_REQUIRE(check_table(layers, n_layers) == 0, "bad layer table (null parameter, or a bare ReLU on the last layer)");
  layout(layers, n_layers, N, nullptr, fwd_floats, bwd_floats, workspace_bytes);
  return CPM_OK;
}

CPM_EXPORT int cpm_layer_chain_forward(const cpm_chain_layer* layers, int n_layers, int N, const float* x,
                                       float* fwd_base, float* y, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  CPM_REQUIRE(layers && n_layers >= 1 && n_layers <= MAX_LAYERS && N >= 1 && x && fwd_base && y, "null chain");
  CPM_REQUIRE(check_table(layers, n_layers) == 0, "bad layer table");
  Off off[MAX_LAYERS];
  layout(layers, n_layers, N, off, nullptr, nullptr, nullptr);
  const float* in = x;
  bool out_filled = false;                                 // the previous layer's GroupNorm left this conv's bias in conv_out
  for (int i = 0; i < n_layers; ++i) {
    const cpm_chain_layer& L = layers[i];
    cpm_conv_desc d = L.conv;
    d.N = N;
    float* conv_out = at(fwd_base, off[i].conv_out, y);
    if (out_filled) cpm_conv_next_output_prepared(1);
    out_filled = false;
    // (a pre-split image serves the bf16x3 arithmetic only: under the other one the float weights are read)
    int rc = (L.w4 && cpm_get_conv_math() == CPM_MATH_BF16X3)
                 ? cpm_conv2d_forward_w4(&d, in, L.w4, nullptr, L.bias, nullptr, 0, L.has_gn ? 0 : L.relu, conv_out, workspace,
                                         workspace_bytes, stream)
                 : cpm_conv2d_forward(&d, in, L.w, nullptr, L.bias, nullptr, 0, L.has_gn ? 0 : L.relu, conv_out, workspace,
                                      workspace_bytes, stream);
    cpm_conv_next_output_prepared(0);
    if (rc != CPM_OK) return rc;
    if (L.has_gn) {
      float* gn_out = at(fwd_base, off[i].gn_out, y);
      // The next conv's output has this GroupNorm's shape and starts from its bias (a bias-only epilogue: a GroupNorm
      // of its own follows, or no ReLU): this kernel writes it on the side -- a reduction-split launch (every grid-head
      // conv on ~100 RoIs) then needs no seed launch (21 per step, 5 us each, between kernels of 50-110 us).
      float* fill = nullptr;
      if (i + 1 < n_layers) {
        const cpm_chain_layer& Nx = layers[i + 1];
        const cpm_conv_desc& e = Nx.conv;
        if (chain_fill() && (Nx.has_gn || !Nx.relu) && e.K == d.K && e.P == d.P && e.Q == d.Q && off[i + 1].conv_out >= 0)
          fill = fwd_base + off[i + 1].conv_out;
      }
      rc = fill ? cpm_groupnorm_forward_fill(conv_out, L.gamma, L.beta, N, n_pix(d), d.K, L.gn_groups, L.eps, L.relu,
                                             gn_out, fwd_base + off[i].mean, fwd_base + off[i].rstd, fill,
                                             layers[i + 1].bias, stream)
                : cpm_groupnorm_forward(conv_out, L.gamma, L.beta, N, n_pix(d), d.K, L.gn_groups, L.eps, L.relu, gn_out,
                                        fwd_base + off[i].mean, fwd_base + off[i].rstd, stream);
      out_filled = fill != nullptr;
      if (rc != CPM_OK) return rc;
      in = gn_out;
    } else {
      in = conv_out;
    }
  }
  return CPM_OK;
}

// dy: gradient at the last layer's output.  Per layer, last to first: GroupNorm(+ReLU) backward into d_conv where the
// layer has one (gamma / beta gradients accumulated into their sinks); the weight + bias gradient on `side_stream`
// (forked from `stream` first; NULL = same stream); the data gradient into d_in (layer 0: into dx, skipped when dx is
// NULL), gated by the layer's INPUT when the previous layer ended in a conv-epilogue ReLU.
CPM_EXPORT int cpm_layer_chain_backward(const cpm_chain_layer* layers, int n_layers, int N, const float* x,
                                        const float* dy, float* fwd_base, float* y, float* bwd_base, float* dx,
                                        void* workspace, size_t workspace_bytes, void* side_workspace,
                                        size_t side_workspace_bytes, void* stream, void* side_stream) {
  CPM_REQUIRE(layers && n_layers >= 1 && n_layers <= MAX_LAYERS && N >= 1 && x && dy && fwd_base && y && bwd_base,
              "null chain");
  CPM_REQUIRE(check_table(layers, n_layers) == 0, "bad layer table");
  Off off[MAX_LAYERS];
  layout(layers, n_layers, N, off, nullptr, nullptr, nullptr);
  const float* g = dy;
  for (int i = n_layers - 1; i >= 0; --i) {
    const cpm_chain_layer& L = layers[i];
    CPM_REQUIRE(L.dw && (!L.has_gn || (L.dgamma && L.dbeta)), "null gradient sink");
    cpm_conv_desc d = L.conv;
    d.N = N;
    const float* in = i == 0 ? x : out_of(layers[i - 1], off[i - 1], fwd_base, y);
    float* d_in = at(bwd_base, off[i].d_in, dx);
    const float* gc = g;                                   // gradient at the conv's own output
    int rc = CPM_OK;
    bool din_cleared = false;
    if (L.has_gn) {
      float* d_conv = bwd_base + off[i].d_conv;
      // the conv's input gradient has this GroupNorm's shape (a 3x3 / stride-1 layer of constant width): cleared on the
      // side for the reduction-split data-gradient launch below (no clear launch of its own)
      const bool same = chain_fill() && d_in && d_in != dx && d.C == d.K && d.H == d.P && d.W == d.Q;
      rc = same ? cpm_groupnorm_backward_zero(g, fwd_base + off[i].conv_out, at(fwd_base, off[i].gn_out, y), L.gamma,
                                              fwd_base + off[i].mean, fwd_base + off[i].rstd, N, n_pix(d), d.K,
                                              L.gn_groups, L.relu, d_conv, L.dgamma, L.dbeta, d_in, stream)
                : cpm_groupnorm_backward(g, fwd_base + off[i].conv_out, at(fwd_base, off[i].gn_out, y), L.gamma,
                                         fwd_base + off[i].mean, fwd_base + off[i].rstd, N, n_pix(d), d.K, L.gn_groups,
                                         L.relu, d_conv, L.dgamma, L.dbeta, stream);
      if (rc != CPM_OK) return rc;
      gc = d_conv;
      din_cleared = same;
    }
    void* ws_w = workspace;
    size_t ws_w_bytes = workspace_bytes;
    void* s_w = stream;
    if (side_stream && side_stream != stream) {
      rc = cpm_stream_fork(stream, side_stream);
      if (rc != CPM_OK) return rc;
      s_w = side_stream; ws_w = side_workspace; ws_w_bytes = side_workspace_bytes;
    }
    if (L.dbias)
      rc = cpm_conv2d_backward_weight_bias(&d, in, gc, L.dw, L.dbias, ws_w, ws_w_bytes, s_w);
    else
      rc = cpm_conv2d_backward_weight(&d, in, gc, L.dw, ws_w, ws_w_bytes, s_w);
    if (rc != CPM_OK) return rc;
    if (d_in) {
      const float* gate = (i > 0 && !layers[i - 1].has_gn && layers[i - 1].relu) ? in : nullptr;
      const cpm_conv_desc dd = L.dgrad_flat ? flat_desc(d) : d;
      if (din_cleared) cpm_conv_next_output_prepared(1);
      if (L.wt && L.wt_w4 && cpm_get_conv_math() == CPM_MATH_BF16X3)
        rc = cpm_conv2d_backward_data_prepared_w4(&dd, gc, L.wt, d_in, 0, nullptr, gate, workspace, workspace_bytes, stream);
      else if (L.wt && !L.wt_w4)
        rc = cpm_conv2d_backward_data_prepared(&dd, gc, L.wt, d_in, 0, nullptr, gate, workspace, workspace_bytes, stream);
      else if (gate)
        rc = cpm_conv2d_backward_data_gated(&dd, gc, L.w, d_in, nullptr, gate, workspace, workspace_bytes, stream);
      else
        rc = cpm_conv2d_backward_data(&dd, gc, L.w, d_in, 0, workspace, workspace_bytes, stream);
      cpm_conv_next_output_prepared(0);
      if (rc != CPM_OK) return rc;
      g = d_in;
    }
  }
  return CPM_OK;
}
