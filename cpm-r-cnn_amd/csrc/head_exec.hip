// Native executor for a stack of [conv3x3 + bias -> GroupNorm -> ReLU] layers (the CMM grid head,
// pet/rcnn/modeling/grid_rcnn/heads/grid_heads.py:41-57,146-152 in the reference: 8 such layers per cascade stage).
//
// With a few dozen RoIs per stage these layers are launch bound: every kernel runs 3-20 us while the Python path
// around it (autograd Function, tensor allocation, ctypes marshalling) costs 25 us per op forward and ~60 us backward.
// The two entry points below run the whole stack from ONE call each -- the same C-ABI kernels in the same order with
// the same arguments as the per-op path (so the results are the same), the host cost of 16 ops (forward) / 16 ops
// (backward) replaced by a loop in C.  Host code only: no kernels of its own.
#include "common.h"

namespace {

inline int n_pix(const cpm_conv_desc& d) { return d.P * d.Q; }
inline size_t al(size_t v) { return (v + 63) / 64 * 64; }          // 256-byte aligned pieces

// Where layer i's tensors live inside the two caller-owned buffers for a batch of N samples (float offsets); -1 = the
// separate tensor (the stack's output y / the input gradient dx).
struct Off { int64_t conv_out, gn_out, mean, rstd, d_conv, d_in; };

constexpr int MAX_LAYERS = 64;

void layout(const cpm_conv_gn_layer* layers, int n_layers, int N, Off* off, size_t* fwd_floats, size_t* bwd_floats,
            size_t* ws_bytes) {
  size_t f = 0, b = 0, ws = 0;
  for (int i = 0; i < n_layers; ++i) {
    cpm_conv_desc d = layers[i].conv;
    d.N = N;
    const size_t out_n = (size_t)N * d.K * d.P * d.Q, in_n = (size_t)N * d.C * d.H * d.W;
    const size_t st = (size_t)N * layers[i].gn_groups;
    Off o;
    o.conv_out = (int64_t)f; f += al(out_n);
    o.gn_out = -1;
    if (i != n_layers - 1) { o.gn_out = (int64_t)f; f += al(out_n); }
    o.mean = (int64_t)f; f += al(st);
    o.rstd = (int64_t)f; f += al(st);
    o.d_conv = (int64_t)b; b += al(out_n);
    o.d_in = -1;
    if (i > 0) { o.d_in = (int64_t)b; b += al(in_n); }
    if (off) off[i] = o;
    if (ws_bytes) {
      const size_t w = cpm_conv2d_workspace_bytes(&d);
      if (w > ws) ws = w;
    }
  }
  if (fwd_floats) *fwd_floats = f;
  if (bwd_floats) *bwd_floats = b;
  if (ws_bytes) *ws_bytes = ws;
}

inline float* at(float* base, int64_t off, float* dflt) { return off >= 0 ? base + off : dflt; }

}  // namespace

CPM_EXPORT int cpm_conv_gn_stack_sizes(const cpm_conv_gn_layer* layers, int n_layers, int N, size_t* fwd_floats,
                                       size_t* bwd_floats, size_t* workspace_bytes) {
  CPM_REQUIRE(layers && n_layers >= 1 && n_layers <= MAX_LAYERS && N >= 1, "1..64 layers, N >= 1");
  layout(layers, n_layers, N, nullptr, fwd_floats, bwd_floats, workspace_bytes);
  return CPM_OK;
}

CPM_EXPORT int cpm_conv_gn_stack_forward(const cpm_conv_gn_layer* layers, int n_layers, int N, const float* x,
                                         float* fwd_base, float* y, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  CPM_REQUIRE(layers && n_layers >= 1 && n_layers <= MAX_LAYERS && N >= 1 && x && fwd_base && y, "null stack");
  Off off[MAX_LAYERS];
  layout(layers, n_layers, N, off, nullptr, nullptr, nullptr);
  const float* in = x;
  for (int i = 0; i < n_layers; ++i) {
    const cpm_conv_gn_layer& L = layers[i];
    CPM_REQUIRE(L.w && L.gamma && L.beta, "null layer parameter");
    cpm_conv_desc d = L.conv;
    d.N = N;
    float* conv_out = fwd_base + off[i].conv_out;
    float* gn_out = at(fwd_base, off[i].gn_out, y);
    int rc = cpm_conv2d_forward(&d, in, L.w, nullptr, L.bias, nullptr, 0, 0, conv_out, workspace, workspace_bytes,
                                stream);
    if (rc != CPM_OK) return rc;
    rc = cpm_groupnorm_forward(conv_out, L.gamma, L.beta, N, n_pix(d), d.K, L.gn_groups, L.eps, 1, gn_out,
                               fwd_base + off[i].mean, fwd_base + off[i].rstd, stream);
    if (rc != CPM_OK) return rc;
    in = gn_out;
  }
  return CPM_OK;
}

// dy: gradient at the last layer's output.  Per layer, last to first: GroupNorm(+ReLU) backward into d_conv (gamma /
// beta gradients accumulated into their sinks), the weight + bias gradient on `side_stream` (forked from `stream`
// first: it needs d_conv; NULL = same stream), the data gradient into d_in (layer 0: into dx, skipped when dx is NULL).
CPM_EXPORT int cpm_conv_gn_stack_backward(const cpm_conv_gn_layer* layers, int n_layers, int N, const float* x,
                                          const float* dy, float* fwd_base, float* y, float* bwd_base, float* dx,
                                          void* workspace, size_t workspace_bytes, void* side_workspace,
                                          size_t side_workspace_bytes, void* stream, void* side_stream) {
  CPM_REQUIRE(layers && n_layers >= 1 && n_layers <= MAX_LAYERS && N >= 1 && x && dy && fwd_base && y && bwd_base,
              "null stack");
  Off off[MAX_LAYERS];
  layout(layers, n_layers, N, off, nullptr, nullptr, nullptr);
  const float* g = dy;
  for (int i = n_layers - 1; i >= 0; --i) {
    const cpm_conv_gn_layer& L = layers[i];
    CPM_REQUIRE(L.dw && L.dgamma && L.dbeta, "null gradient sink");
    cpm_conv_desc d = L.conv;
    d.N = N;
    const float* in = i == 0 ? x : at(fwd_base, off[i - 1].gn_out, y);
    float* d_conv = bwd_base + off[i].d_conv;
    float* d_in = at(bwd_base, off[i].d_in, dx);
    int rc = cpm_groupnorm_backward(g, fwd_base + off[i].conv_out, at(fwd_base, off[i].gn_out, y), L.gamma,
                                    fwd_base + off[i].mean, fwd_base + off[i].rstd, N, n_pix(d), d.K, L.gn_groups, 1,
                                    d_conv, L.dgamma, L.dbeta, stream);
    if (rc != CPM_OK) return rc;
    void* ws_w = workspace;
    size_t ws_w_bytes = workspace_bytes;
    void* s_w = stream;
    if (side_stream && side_stream != stream) {
      rc = cpm_stream_fork(stream, side_stream);
      if (rc != CPM_OK) return rc;
      s_w = side_stream; ws_w = side_workspace; ws_w_bytes = side_workspace_bytes;
    }
    if (L.dbias)
      rc = cpm_conv2d_backward_weight_bias(&d, in, d_conv, L.dw, L.dbias, ws_w, ws_w_bytes, s_w);
    else
      rc = cpm_conv2d_backward_weight(&d, in, d_conv, L.dw, ws_w, ws_w_bytes, s_w);
    if (rc != CPM_OK) return rc;
    if (d_in) {
      if (L.wt)
        rc = cpm_conv2d_backward_data_prepared(&d, d_conv, L.wt, d_in, 0, nullptr, nullptr, workspace, workspace_bytes,
                                               stream);
      else
        rc = cpm_conv2d_backward_data(&d, d_conv, L.w, d_in, 0, workspace, workspace_bytes, stream);
      if (rc != CPM_OK) return rc;
      g = d_in;
    }
  }
  return CPM_OK;
}
