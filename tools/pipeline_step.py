"""C2 step with the prefetch pipeline (the reference's PrefetchTrainPipelineSparseDist order: the index stage of batch k + 1
is issued on a side stream before the backward of batch k) against the serial step, same module, same batches.
    python tools/pipeline_step.py [--steps 200]
MI355_PREFETCH_C=0 keeps the pinning prefetch (per-slot counters + ref-counter atomics) for comparison."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--rows", type=int, default=10_000_000)
a = ap.parse_args()
dev = torch.device("cuda")
batches = bench.zipf_batches(a.rows, 0.99, a.batch, 8, dev)
module = bench.build_module(a.rows, 128, dev)
module.train()
grad = (torch.randn(a.batch, 128, device=dev) * 0.01).to(torch.bfloat16)
with torch.no_grad():
    for k, o in batches:
        module._forward_impl(k, o, train=True)
nb = len(batches)


def serial(steps):
    for i in range(steps):
        k, o = batches[i % nb]
        out = module(k, o)
        out.backward(grad)


print("serial    %.4f ms/step" % bench.timed_loop(serial, a.steps))
print("pipelined %.4f ms/step  (staged on the partitioned path: %s)" % (bench.pipelined_ms(module, batches, grad, a.steps), module._pf_c_used))
print("serial    %.4f ms/step" % bench.timed_loop(serial, a.steps))
