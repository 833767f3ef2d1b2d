import sys, time, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recsys-examples_amd")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
batches = bench.zipf_batches(10_000_000, 0.99, 65536, 60, dev)
m = bench.build_module(10_000_000, 128, dev); m.train()
with torch.no_grad():
    for k, o in batches: m._forward_impl(k, o, train=True)
grad = (torch.randn(65536, 128, device=dev) * 0.01).bfloat16()
# tiny batch: GPU work is negligible, the loop time is the host cost of a step
tk, to = batches[0][0][:64].contiguous(), torch.arange(0, 65, dtype=torch.int64, device=dev)
tg = grad[:64].contiguous()
for _ in range(20):
    out, st = m._forward_impl(tk, to, train=True); m._backward_impl(st, tg)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300):
    out, st = m._forward_impl(tk, to, train=True); m._backward_impl(st, tg)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue per step {1e6*(t1-t0)/300:.1f} us; incl. drain {1e6*(t2-t0)/300:.1f} us")
