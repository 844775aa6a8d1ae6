import ctypes, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cotnet_amd import _lib
L = _lib.lib(); P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda"; N, H = 80, 224
x = torch.randn(N, 3, H, H, device=dev); w = torch.randn(64, 3, 7, 7, device=dev) / 12
y = torch.empty(N, 64, 112, 112, device=dev); gy = torch.randn_like(y); gw = torch.empty_like(w)
ws = torch.empty(int(L.cot_stem7x7s2_workspace(N, H, H)), dtype=torch.uint8, device=dev)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("fp32 stem fwd us", t(lambda: L.cot_stem7x7s2_forward(P(x), P(w), P(y), N, H, H, 0, None)))
print("fp32 stem wgrad us", t(lambda: L.cot_stem7x7s2_backward_weight(P(gy), P(x), P(gw), P(ws), N, H, H, 0, None)))
conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev)
xr = x.clone()
print("torch fwd us", t(lambda: conv(xr)))
yy = conv(xr)
def bw():
    conv.weight.grad = None
    yy2 = conv(xr); yy2.backward(gy)
print("torch fwd+bwd us", t(bw))
