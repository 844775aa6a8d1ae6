"""eager vs HIP-graph gradients on identical state (lr = 0 so nothing moves)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cotnet_amd
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
from cotnet_amd.graph_step import GraphedTrainStep

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.backends.cudnn.deterministic = os.environ.get("DET", "0") == "1"
print("deterministic", torch.backends.cudnn.deterministic)
torch.manual_seed(0)
B = int(os.environ.get("B", "16"))
model = to_mixed_bf16(cotnet_amd.create_model("cotnet50", num_classes=1000).to(dev)).train()
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.momentum = 0.0  # freeze running stats so repeated passes are identical
opt = FlatSGD(model, lr=0.0, momentum=0.0, weight_decay=0.0)
x = torch.randn(B, 3, 224, 224, device=dev).bfloat16()
t = torch.randint(0, 1000, (B,), device=dev)
lf = lambda o, tt: torch.nn.functional.cross_entropy(o.float(), tt)

def eager():
    opt.zero_grad()
    loss = lf(model(x), t)
    loss.backward()
    opt.reducer.finish()
    return loss.item(), [b.flat.float().clone() for b in opt.reducer.buckets]

l1, g1 = eager()
l2, g2 = eager()
print("eager loss", l1, l2, "eager-vs-eager max diff", [float((a - b).abs().max()) for a, b in zip(g1, g2)])
gs = GraphedTrainStep(model, opt, lf, x, t)
for rep in range(3):
    for b in opt.reducer.buckets:
        b.flat.fill_(float("nan"))
    loss = gs()
    torch.cuda.synchronize()
    g3 = [b.flat.float().clone() for b in opt.reducer.buckets]
    print(f"replay {rep}: loss", float(loss), "graph-vs-eager max diff", [float((a - b).abs().max()) for a, b in zip(g1, g3)],
          "nan counts", [int(torch.isnan(a).sum()) for a in g3], "sizes", [a.numel() for a in g3])
# which parameters differ most
names = {p: n for n, p in model.named_parameters()}
for b, a, c in zip(opt.reducer.buckets, g1, g3):
    off = 0
    al = max(1, 16 // b.flat.element_size())
    worst = []
    for p in b.params:
        n = p.numel()
        d = (a[off:off + n] - c[off:off + n]).abs().max().item() if n else 0.0
        if not (d < 1e-2):
            worst.append((names[p], d, float(a[off:off + n].abs().max())))
        off += (n + al - 1) // al * al
    print("bucket", b.key, b.flat.dtype, "params differing:", len(worst), worst[:6])
