"""The forced-sharded (W = 1) C2 step in its overlapped schedule, N steps and nothing else: for a kernel trace / timeline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from dynamicemb.sharded import ShardedPooledLookup
batches = bench.zipf_batches(10_000_000, 0.99, 65536, 40, dev)
sh = ShardedPooledLookup(10_000_000, 128, dev, 1, 0, mode="partial", keys_per_step=int(65536 * 5.5), batch=65536)
grad = (torch.randn(65536, 128, device=dev) * 0.01).to(torch.bfloat16)
with torch.no_grad():
    for k, o in batches: sh.forward(k, o)
def run(n):
    for i in range(n):
        k, o = batches[i % 40]
        out, st = sh.forward(k, o, next_batch=batches[(i + 1) % 40])
        sh.backward(st, grad)
run(40); torch.cuda.synchronize()
t0 = time.perf_counter(); run(200); torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 200 * 1e3, "mode", sh.mode)
dist.destroy_process_group()
