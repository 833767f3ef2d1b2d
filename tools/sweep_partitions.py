# threshold sweep of the partitioned index stage: C2-like steps at several batch sizes (run with MI355_FUSED_PART=0 / 1)
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extended as be
import torch
torch.cuda.set_device(0)
out = {}
for B in (12288, 24576, 49152, 98304):
    r = be.run_table_case(f"b{B}", 10_000_000, 128, B, 40)
    out[B] = (round(r["ms_per_step"] * 1000, 1), int(r["keys_per_step"]))
r = be.run_table_case("c2s", 10_000_000, 128, 65536, 40, hotness1=True, pooling="NONE")
out["seq65k"] = round(r["ms_per_step"] * 1000, 1)
print(os.environ.get("MI355_FUSED_PART", "1"), json.dumps(out))
