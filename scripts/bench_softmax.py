import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotnet_amd.aggregation_zeropad import aggregation_zeropad, aggregation_zeropad_softmax
dev = "cuda"
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(80, 64, 56, 56, device=dev, dtype=dt).requires_grad_(True)
    lg = torch.randn(80, 1, 8, 9, 56, 56, device=dev, dtype=dt).requires_grad_(True)
    go = torch.randn(80, 64, 56, 56, device=dev, dtype=dt)
    def fused():
        y = aggregation_zeropad_softmax(x, lg, 3, 1, 1, 1); torch.autograd.grad(y, (x, lg), go)
    def composed():
        y = aggregation_zeropad(x, torch.softmax(lg, dim=3), 3, 1, 1, 1); torch.autograd.grad(y, (x, lg), go)
    for name, f in (("fused", fused), ("composed", composed)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"{str(dt):15s} softmax+aggregation fwd+bwd N80xC64x56x56 {name:9s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
