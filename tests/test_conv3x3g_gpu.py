"""The grouped 3x3 MFMA convolution kernels (csrc/conv3x3g.hip behind cot_conv3x3g_*, opt-in COT_CONV3X3=hip) on the GPU
against torch's convolution evaluated in fp32 on the same bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from cotnet_amd import conv3x3g as c3

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(a, b, rel):
    return ((a.float() - b).abs() <= rel * (b.abs() + b.abs().mean())).all()


# (N, C, groups, H): CotLayer.key_embed at the four CoTNet-50 stages (models/cotnet.py:43-47), CoXtLayer's groups=8
# (:112-116), the ungrouped SplitAttn 3x3 of SE-CoTNetD, and shapes where nothing is aligned
CASES = [(4, 64, 4, 56), (4, 128, 4, 28), (3, 256, 4, 14), (3, 512, 4, 7), (2, 192, 8, 28), (2, 64, 1, 40),
         (2, 32, 4, 9), (1, 16, 2, 3), (1, 8, 1, 1)]


@pytest.mark.parametrize("N,C,G,H", CASES)
def test_matches_torch_convolution(N, C, G, H, monkeypatch):
    monkeypatch.setattr(c3, "MODE", "hip")
    torch.manual_seed(C + H)
    conv = nn.Conv2d(C, C, 3, padding=1, groups=G, bias=False).to(DEV).bfloat16()
    x = torch.randn(N, C, H, H, device=DEV).bfloat16().requires_grad_(True)
    gy = torch.randn(N, C, H, H, device=DEV).bfloat16()
    assert c3.eligible(conv, x)
    y = c3.conv3x3(conv, x)
    y.backward(gy)
    torch.cuda.synchronize()
    xr = x.detach().float().requires_grad_(True)
    wr = conv.weight.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1, 1, G)
    yr.backward(gy.float())
    assert _close(y, yr.detach(), 1e-2)
    assert _close(x.grad, xr.grad, 1e-2)
    assert _close(conv.weight.grad, wr.grad, 1e-2)


def test_first_and_last_waves_do_not_read_outside_the_tensor(monkeypatch):
    """x is a view placed at the very start / end of its allocation neighbours: NaN-filled guard tensors allocated right
    before and after must not leak into the result (the first / last waves take the bounds-checked path)"""
    monkeypatch.setattr(c3, "MODE", "hip")
    torch.manual_seed(0)
    N, C, G, H = 2, 32, 4, 12
    buf = torch.full((3, N, C, H, H), float("nan"), device=DEV).bfloat16()
    buf[1] = torch.randn(N, C, H, H, device=DEV).bfloat16()
    x = buf[1]
    conv = nn.Conv2d(C, C, 3, padding=1, groups=G, bias=False).to(DEV).bfloat16()
    y = c3.conv3x3(conv, x)
    yr = F.conv2d(x.float(), conv.weight.float(), None, 1, 1, 1, G)
    assert not torch.isnan(y.float()).any()
    assert _close(y, yr, 1e-2)


def test_weight_gradient_is_deterministic(monkeypatch):
    monkeypatch.setattr(c3, "MODE", "hip")
    torch.manual_seed(0)
    conv = nn.Conv2d(64, 64, 3, padding=1, groups=4, bias=False).to(DEV).bfloat16()
    x = torch.randn(16, 64, 56, 56, device=DEV).bfloat16()
    gy = torch.randn(16, 64, 56, 56, device=DEV).bfloat16()
    grads = []
    for _ in range(3):
        conv.weight.grad = None
        c3.conv3x3(conv, x).backward(gy)
        grads.append(conv.weight.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


@pytest.mark.parametrize("N,C,G,H", [(80, 64, 4, 56), (80, 128, 4, 28), (80, 256, 4, 14), (80, 512, 4, 7), (5, 64, 4, 12), (3, 128, 1, 9)])
def test_weight_gradient_lds_staged_kernel(N, C, G, H):
    """cot_conv3x3g_backward_weight_guarded (csrc/conv_wgrad2.hip, TAPS form) at the batch the benchmark runs and at ragged
    sizes: x lives inside an allocation whose margins are NaN (what the shifted 16-byte copies read there must be cleared by
    selection); against torch's conv2d weight gradient in fp32 on the same operands; twice -> same bits; and the kernel that ran
    is the LDS-staged one"""
    import ctypes
    from cotnet_amd import _lib
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(C + H)
    HW, lead = H * H, (H + 1 + 7) // 8 * 8
    flat = torch.full((N * C * HW + 2 * lead,), float("nan"), device=DEV).bfloat16()
    x = flat[lead:lead + N * C * HW].view(N, C, H, H)
    x.copy_(torch.randn(N, C, H, H, device=DEV))
    gy = torch.randn(N, C, H, H, device=DEV).bfloat16()
    wf = torch.zeros(C, C // G, 3, 3, device=DEV, requires_grad=True)
    F.conv2d(x.float(), wf, None, 1, 1, 1, G).backward(gy.float())
    masks = torch.empty(int(L.cot_conv3x3g_masks_bytes(H, H)), dtype=torch.uint8, device=DEV)
    assert L.cot_conv3x3g_masks(P(masks), H, H, st) == 0
    ws = torch.empty(int(L.cot_conv3x3g_workspace(N, C, C, G, H, H)), dtype=torch.uint8, device=DEV)
    outs = []
    for _ in range(2):
        gw = torch.full((C, C // G, 3, 3), float("nan"), device=DEV).bfloat16()
        ws.fill_(0x7f)
        rc = L.cot_conv3x3g_backward_weight_guarded(P(gy), P(x), P(gw), P(masks), P(ws), N, C, C, G, H, H, _lib.COT_BF16, lead, st)
        assert rc == 0, L.cot_last_error()
        torch.cuda.synchronize()
        outs.append(gw)
    try:  # which kernels the call takes: a dry run (nothing is launched) fills the launch log
        assert L.cot_set_tuning(26, 1) == 0
        assert L.cot_conv3x3g_backward_weight_guarded(P(gy), P(x), P(gw), P(masks), P(ws), N, C, C, G, H, H, _lib.COT_BF16, lead, st) == 0
        buf = ctypes.create_string_buffer(4096)
        L.cot_launch_log(buf, 4096)
    finally:
        assert L.cot_set_tuning(26, 0) == 0
    assert b"conv1x1_wgrad_lds2<9, 1" in buf.value and b"block 832" in buf.value, buf.value
    assert torch.equal(outs[0], outs[1])
    scale = wf.grad.abs().max().item()
    assert (outs[0].float() - wf.grad).abs().max().item() <= 1e-2 * scale


def test_cot_layer_with_all_hip_convolutions():
    """CotLayer forward+backward with every convolution on the hand-written kernels: not further from the fp32 truth
    than the default (MIOpen) path (tests/truth.py)"""
    from cotnet_amd.cotnet import CotLayer
    from cotnet_amd.flat_sgd import to_mixed_bf16
    from tests import truth
    torch.manual_seed(1)
    layer = to_mixed_bf16(CotLayer(64, 3).to(DEV)).train()
    x = torch.randn(4, 64, 28, 28, device=DEV).bfloat16()
    gy = torch.randn(4, 64, 28, 28, device=DEV).bfloat16()
    truth.check_against_truth(layer, x, gy, cand=dict(truth.ROUND1, conv1x1="hip", conv3x3="hip"))
