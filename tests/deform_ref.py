"""An independent, differentiable formulation of deformable conv v1 in plain torch (float64, CPU): bilinear
sampling written with gather + arithmetic so autograd supplies d input, d offset and d weight.  Used to check the
C oracle's hand-derived gradients (oracle/cpm_oracle.c: orc_deform_conv) and, through it, the HIP kernels."""
import torch


def deform_conv_torch(x, offset, weight, stride=1, pad=1, dil=1, groups=1, dg=1):
    N, C, H, W = x.shape
    K, Cg, R, S = weight.shape
    P = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Q = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    dt = x.dtype
    pp = torch.arange(P, dtype=dt).view(1, 1, P, 1) * stride - pad
    qq = torch.arange(Q, dtype=dt).view(1, 1, 1, Q) * stride - pad
    off = offset.view(N, dg, R * S, 2, P, Q)
    cols = []
    xf = x.reshape(N, dg, C // dg, H * W)
    for t in range(R * S):
        i, j = divmod(t, S)
        h = pp + i * dil + off[:, :, t, 0]            # [N, dg, P, Q]
        w = qq + j * dil + off[:, :, t, 1]
        ok = ((h > -1) & (w > -1) & (h < H) & (w < W)).to(dt)
        h0, w0 = torch.floor(h.detach()), torch.floor(w.detach())
        lh, lw = h - h0, w - w0
        val = 0
        for dh, dw_, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)),
                            (1, 1, lh * lw)):
            hi, wi = h0 + dh, w0 + dw_
            inb = ((hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1)).to(dt)
            idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).long().view(N, dg, 1, P * Q)
            v = torch.gather(xf, 3, idx.expand(N, dg, C // dg, P * Q)).view(N, dg, C // dg, P, Q)
            val = val + v * (wt * inb * ok).unsqueeze(2)
        cols.append(val.reshape(N, C, P, Q))
    col = torch.stack(cols, 2)                         # [N, C, RS, P, Q]
    col = col.view(N, groups, C // groups, R * S, P * Q)
    wg = weight.view(groups, K // groups, Cg, R * S)
    y = torch.einsum("ngctm,gkct->ngkm", col, wg)
    return y.reshape(N, K, P, Q)
