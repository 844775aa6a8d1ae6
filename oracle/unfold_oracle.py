"""The analytic oracle the reference's own self-tests use -- TEST INFRASTRUCTURE.

`aggregation_unfold` is the nn.Unfold + broadcast-multiply + sum formula from
cupy_layers/aggregation_zeropad.py:247-251 (and :366 of the mix file for the
two-kernel variant).  It is differentiable through torch autograd, which gives
the reference gradients the self-tests compare against (:254-260).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (cotnet_amd/) never does.
"""
import torch
from torch.nn.modules.utils import _pair


def out_hw(H, W, kernel_size, stride, padding, dilation):
    k, s, p, d = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    # aggregation_zeropad.py:120-121
    Ho = int((H + 2 * p[0] - (d[0] * (k[0] - 1) + 1)) / s[0] + 1)
    Wo = int((W + 2 * p[1] - (d[1] * (k[1] - 1) + 1)) / s[1] + 1)
    return Ho, Wo


def aggregation_unfold(x, w, kernel_size=3, stride=1, padding=0, dilation=1):
    """x [N,C,H,W], w [N,heads,wC,k*k,Ho,Wo] -> [N,heads*C,Ho,Wo] (head-major)."""
    n, c_x, H, W = x.shape
    _, heads, c_w, taps, _, _ = w.shape
    Ho, Wo = out_hw(H, W, kernel_size, stride, padding, dilation)
    unfold = torch.nn.Unfold(kernel_size=kernel_size, dilation=dilation, padding=padding, stride=stride)
    x2 = unfold(x).view(n, c_x // c_w, c_w, taps, Ho, Wo)
    w = w.reshape(n, heads, c_w, taps, Ho, Wo)
    return (w.unsqueeze(2) * x2.unsqueeze(1)).sum(-3).view(n, heads * c_x, Ho, Wo)


def aggregation_mix_unfold(x, w1, w2, stride=1, padding1=1, padding2=2, dilation=1):
    """3x3 and 5x5 aggregation of the same input, concatenated on dim 1 (mix.py:366)."""
    y1 = aggregation_unfold(x, w1, 3, stride, padding1, dilation)
    y2 = aggregation_unfold(x, w2, 5, stride, padding2, dilation)
    return torch.cat([y1, y2], dim=1)
