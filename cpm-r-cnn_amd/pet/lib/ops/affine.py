"""AffineChannel2d (counterpart of pet/lib/ops/affine.py:5-17): a frozen per-channel scale/shift.
On the hot path it is never run as its own kernel: the conv that feeds it applies weight/bias in its epilogue
(see pet.lib.ops.conv.conv2d); calling the module directly falls back to torch broadcasting arithmetic."""
import torch
import torch.nn as nn


class AffineChannel2d(nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.empty(num_features).uniform_())
        self.bias = nn.Parameter(torch.zeros(num_features))

    def forward(self, x):
        return x * self.weight.view(1, self.num_features, 1, 1) + self.bias.view(1, self.num_features, 1, 1)
