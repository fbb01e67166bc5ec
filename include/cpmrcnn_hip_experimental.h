/* Experimental entry points of libcpmrcnn_hip.so: built, parity-tested and measured, NOT on the product path (no
 * reference-side binding needs them; DESIGN.md section 8 has the measurements).  Kept so that the experiments stay
 * reproducible: tools/bench_conv.py --math sp, tools/ring_*.sh, tests/test_gpu_conv_sp.py. */
#ifndef CPMRCNN_HIP_EXPERIMENTAL_H
#define CPMRCNN_HIP_EXPERIMENTAL_H
#include "cpmrcnn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- split-plane ("SP") operands for the CPM_MATH_BF16X3 arithmetic ------------------------------------------------
 * The 3-term split-bf16 product needs hi = bf16(v), lo = bf16(v - hi) of every operand.  The plain entry points above
 * split fp32 operands inside the kernel, for every tile that reads them; the _sp entry points take the operands
 * ALREADY split, as bf16 blocks in memory, and stage them by LDS-DMA through a ring of stages (conv_sp.hip): same
 * arithmetic, same results to rounding of the accumulation order, no conversion work in the loop.
 *   SP of a [rows][C] fp32 matrix (rows = NHWC pixels, or (k, r, s) rows of a KRSC weight) = [rows][C/32][2][32] bf16:
 *   per row and 32-channel block 32 hi values then 32 lo values (128 bytes: one cache line per row and reduction
 *   step) -- 4*C bytes, like the fp32 row.  C % 32 == 0, 128-byte aligned.
 * cpm_split_planes makes one; the convolutions can also emit their result in that form (y_sp / dx_sp, may be NULL)
 * for the convolution that consumes it.  The fp32 tensors stay the interface (x, w are still required): an operand
 * without an SP twin (x_sp / w_sp NULL), or a shape outside the DMA kernel's rules (C/groups % 32, K/groups > 32),
 * takes the in-kernel split path.  Replaces the same ATen/cuDNN calls as cpm_conv2d_* (pet/lib/ops call sites above). */
int cpm_split_planes(const float* x, int64_t rows, int channels, void* sp, void* stream);
int cpm_conv2d_forward_sp(const cpm_conv_desc* d, const float* x, const void* x_sp, const float* w, const void* w_sp,
                          const float* scale, const float* shift, const float* residual, int res_mode, int relu,
                          float* y, void* y_sp, void* stream);
/* wt / wt_sp: the prepared data-gradient weight image (cpm_weights_to_dgrad_batched) and its SP twin */
int cpm_conv2d_backward_data_sp(const cpm_conv_desc* d, const float* dy, const void* dy_sp, const float* wt,
                                const void* wt_sp, float* dx, void* dx_sp, int accumulate, const float* in_scale,
                                const float* in_act, void* stream);

#ifdef __cplusplus
}
#endif
#endif
