"""helper for tests/test_bench_selection_cpu.py: the way bench.py shares rank 0's kernel-set verdict with the other ranks
(rendezvous store: set / wait / get), run with world size 2 on gloo"""
import datetime
import os
import time

import torch.distributed as dist

rank = int(os.environ["RANK"])
dist.init_process_group("gloo", init_method="env://", timeout=datetime.timedelta(minutes=5))
store = dist.distributed_c10d._get_default_store()
if rank == 0:
    time.sleep(1.0)  # "probing"
    chosen = "new"
    store.set("cot_kernel_set", chosen)
else:
    store.wait(["cot_kernel_set"], datetime.timedelta(minutes=2))
    chosen = store.get("cot_kernel_set").decode()
print(f"VERDICT rank{rank} {chosen}", flush=True)
dist.barrier()
dist.destroy_process_group()
