"""C2 step with the prefetch pipeline (the reference's PrefetchTrainPipelineSparseDist order: the index stage of batch k + 1
is issued on a side stream before the backward of batch k) against the serial step, same module, same batches.
    python tools/pipeline_step.py [--steps 200]"""
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
        out, st = module._forward_impl(k, o, train=True)
        module._backward_impl(st, grad)


def pipelined(steps, side):
    main = torch.cuda.current_stream()
    module.prefetch(*batches[0])
    for i in range(steps):
        k, o = batches[i % nb]
        out, st = module._forward_impl(k, o, train=True)      # consumes the prefetched state: gather only
        side.wait_stream(main)                                 # the side stream sees the table as of this point
        with torch.cuda.stream(side):
            module.prefetch(*batches[(i + 1) % nb])
        module._backward_impl(st, grad)
    # drain the last prefetched state
    out, st = module._forward_impl(*batches[steps % nb], train=True)
    module._backward_impl(st, grad)


def timeit(fn, steps, *args):
    fn(20, *args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps, *args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("serial    %.4f ms/step" % timeit(serial, a.steps))
side = torch.cuda.Stream()
print("pipelined %.4f ms/step" % timeit(pipelined, a.steps, side))
print("serial    %.4f ms/step" % timeit(serial, a.steps))
