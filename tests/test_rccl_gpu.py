"""The data-parallel gradient path on a REAL RCCL communicator (backend "nccl" = RCCL on ROCm), world size 1 -- all a
1-GPU box offers.  A one-rank all-reduce(AVG) is the identity, so the gradients and the SGD trajectory must equal the
no-communication path bit for bit, while the code that runs is the multi-GPU one: flat bf16 / fp32 buckets, event
recorded on the compute stream, ncclAllReduce enqueued on the side stream, stream-wait before the optimizer kernels, flat
broadcast of the initial state, distribute_bn (reference: train.py:112-115, utils/distributed.py:57-67)."""
import copy
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def rccl_world_of_one():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    f = tempfile.NamedTemporaryFile(delete=False)
    f.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{f.name}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()
    if os.path.exists(f.name):  # (the file store removes its file itself)
        os.unlink(f.name)


def _net():
    from cotnet_amd.cotnet import Bottleneck
    return nn.Sequential(nn.Conv2d(3, 256, 3, padding=1, bias=False), Bottleneck(256, 64), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
                         nn.Linear(256, 10))


def test_flat_sgd_over_rccl_equals_the_local_path(rccl_world_of_one):
    from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
    torch.manual_seed(0)
    a = to_mixed_bf16(_net().to(DEV)).train()
    b = copy.deepcopy(a)
    oa = FlatSGD(a, lr=0.05, weight_decay=1e-4, bucket_mb=0.25, force_collectives=True)   # several buckets
    ob = FlatSGD(b, lr=0.05, weight_decay=1e-4, bucket_mb=0.25)
    assert oa.reducer.enabled and oa.reducer.comm_stream is not None and oa.reducer._avg_op
    assert not ob.reducer.enabled and len(oa.reducer.buckets) > 2
    x = torch.randn(8, 3, 16, 16, device=DEV).bfloat16()
    t = torch.randint(0, 10, (8,), device=DEV)
    for _ in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            nn.functional.cross_entropy(m(x).float(), t).backward()
            o.step()
    torch.cuda.synchronize()
    assert oa.reducer._launched == 3 * len(oa.reducer.buckets)
    # (MIOpen's weight-gradient kernels are not bit-reproducible from run to run, so the two replicas are compared to
    # bf16 rounding; a missing stream wait or a wrong AVG shows up as zeros / garbage / a factor, not as an ulp)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.isfinite(pa.float()).all(), n
        assert (pa.float() - pb.float()).abs().max() <= 2e-2 * pb.float().abs().max() + 1e-3, n


def test_distribute_bn_and_state_broadcast_over_rccl(rccl_world_of_one):
    from cotnet_amd.data_parallel import GradBucketReducer, distribute_bn
    torch.manual_seed(1)
    m = _net().to(DEV)
    before = {n: b.clone() for n, b in m.named_buffers()}
    distribute_bn(m, reduce=True)   # world of one: returns early, but must not throw with a live nccl group
    r = GradBucketReducer(m, force_collectives=True)   # broadcasts parameters + buffers in flat messages
    torch.cuda.synchronize()
    for n, b in m.named_buffers():
        assert torch.equal(b, before[n]), n
    r.remove()
