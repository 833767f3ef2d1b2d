import os, sys, socket
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MI355_ENV_LIVE", "1")
import numpy as np, torch
import torch.distributed as dist
import test_twin_gpu as T
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
T.test_pre_communication_dedup_against_the_dict_twin(dist.group.WORLD)
print("pre_comm ok")
T.test_table_grows_under_the_prefetch_pipeline_order()
print("pipeline ok")
from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                          DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)
dims, fmap = [8, 8], [0, 1]
opts = [DynamicEmbTableOptions(dim=d, init_capacity=256, max_capacity=16384, max_load_factor=0.5, index_type=torch.int64,
                               embedding_dtype=torch.float32, score_strategy=DynamicEmbScoreStrategy.STEP,
                               initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG)) for d in dims]
m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=fmap, pooling_mode=DynamicEmbPoolingMode.SUM,
                                    output_dtype=torch.float32, optimizer=EmbOptimType.ADAM, learning_rate=0.05, device=torch.device("cuda", 0))
m.train()
torch.cuda.synchronize()
print("fresh module: size", m.size(), "bucket_sizes", m.table.bucket_sizes.tolist())
st = m.table.table_storage_
C = 128
nb = m.table.num_buckets_
raw = st.cpu().numpy()
stride = (9 + 8) * C
nonempty = 0
for b in range(nb):
    ks = raw[b * stride: b * stride + 8 * C].view(np.uint64)
    ne = ks[(ks & np.uint64(0xFFFFFFFFFFFFFFFC)) != np.uint64(0xFFFFFFFFFFFFFFFC)]
    nonempty += ne.size
    if ne.size: print("bucket", b, "stale keys", ne[:10])
print("non-empty key words in a fresh table:", nonempty)
for t in range(2):
    f, r = m.lookup_rows(torch.tensor([123, 364, 87], device="cuda"), t)
    print("table", t, "found", f.tolist())
print("values ptrs", [v.data_ptr() for v in m.values], "row sample", m.values[0][:2, :4].tolist())
rng = np.random.default_rng(12)
keys, off = T._batch(rng, 2, 64, 200)
out, st_ = m._forward_impl(torch.from_numpy(keys).to("cuda"), torch.from_numpy(off).to("cuda"), train=True)
from oracle.dict_twin import DictEmbeddingTwin
torch.cuda.synchronize()
twin = DictEmbeddingTwin(dims, fmap, "SUM", "adam", lr=0.05)
ref = twin.forward(keys, off, True)
o = out.double().cpu().numpy()
bad = np.argwhere(np.abs(o - ref).max(1) > 1e-3).ravel()
print("bad bags", bad[:20], "of", o.shape)
F, B = 2, 64
for bb in bad[:6]:
    for f in range(F):
        lo, hi = off[f * B + bb], off[f * B + bb + 1]
        ks = keys[lo:hi]
        fnd, rows = m.lookup_rows(torch.from_numpy(ks).to("cuda"), f)
        print("bag", bb, "feature", f, "keys", ks.tolist(), "found", fnd.tolist(), "row0", rows[:, 0].tolist(), "out", o[bb, 8 * f], "ref", ref[bb, 8 * f])
print("size after", m.size(), "uoff", st_.uoff.tolist())
print("slots", st_.slots[: int(st_.uoff[-1])].tolist()[:40])
ra = st_.row_addr[: int(st_.uoff[-1])]
print("row_addr min/max", int(ra.min()), int(ra.max()), "values ptrs", [v.data_ptr() for v in m.values], "caps", m.table.per_table_capacity_)
