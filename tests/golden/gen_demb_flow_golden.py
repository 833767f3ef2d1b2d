"""Generates tests/golden/demb_flow_golden.npz: more of the path-A oracle pinned on the reference's own PURE-PYTHON code,
pulled out of its AST (the modules themselves cannot be imported: they need the CUDA extension / torchrec / fbgemm_gpu).

1. `LinearBucketTable._bucketize_and_pad` + `_deterministic_insert` (corelib/dynamicemb/dynamicemb/scored_hashtable.py
   :1451-1558): the reference's DEMB_DETERMINISM_MODE wave construction, executed here with its three native calls
   (`bucketize_keys`, `table_insert`, `table_lookup`) bound to the CPU oracle (oracle/demb_oracle.c).  Fixture = key
   streams + the slot index of every key + the table arena after the insert.  It pins the wave ORDER (which key of a
   bucket goes into which launch, padding, filtering) that the oracle's own `insert_deterministic` and the package's
   `LinearBucketTable._deterministic_insert` restate -- on top of the oracle's probe / eviction, which stay a restatement.
2. `get_optimizer_state_dim` / `get_optimizer_ckpt_state_dim` (dynamicemb/optimizer.py:36-75): row layout [emb | state].
3. `zipf` (benchmark/dataset_generator.py:75-103): the benchmark key stream, run on the CPU under a fixed torch seed.
4. the planner's capacity arithmetic (dynamicemb_config.py:661-760): table alignment, per-rank bucket layout, HBM-budget
   capacity.
5. the key -> owner routing rule of the row-wise exchange (the reference's CPU check of its bucketize kernel,
   test_hash_roundrobin_kuairand.py:16-45).

Run in the build container only:   python tests/golden/gen_demb_flow_golden.py
"""
import ast
import enum
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

REF = "/root/reference/corelib/dynamicemb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demb_flow_golden.npz")


def functions_of(path, names, cls=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    return [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]


# ------------------------------------------------------------------------------------------------ 1. deterministic insert
class ScorePolicy(enum.IntEnum):
    CONST = 0
    ASSIGN = 1
    ACCUMULATE = 2
    GLOBAL_TIMER = 3
    LRU_LFU = 4


class Stub:
    """what the two extracted methods touch of `self`"""

    def __init__(self, capacities, C):
        self.t = orc.OracleTable(capacities, C)
        self.table_storage_ = "storage"
        self.table_bucket_offsets_ = torch.from_numpy(self.t.tbo)
        self.bucket_capacity_ = self.t.C
        self.bucket_sizes = "sizes"
        self._ref_counter = "counter"
        self.num_buckets_ = self.t.num_buckets

    def bucketize_keys(self, keys, table_ids):
        ko, off, inv = self.t.bucketize(keys.numpy().view(np.uint64), table_ids.numpy())
        return torch.from_numpy(ko.view(np.int64)), torch.from_numpy(off), torch.from_numpy(inv)


def make_ns(stub):
    def table_insert(storage, tbo, C, sizes, keys, tids, scores, policy, counter):
        s = None if scores is None else scores.view(torch.int64).numpy().view(np.uint64)
        stub.t.insert(keys.numpy().view(np.uint64), tids.numpy(), s, int(policy), timer=0)

    def table_lookup(storage, tbo, C, keys, tids, scores, policy):
        so, fo, idx = stub.t.lookup(keys.numpy().view(np.uint64), tids.numpy(), None, int(policy))
        return torch.from_numpy(so), torch.from_numpy(fo), torch.from_numpy(idx)

    from typing import Optional, Tuple
    return dict(torch=torch, Optional=Optional, Tuple=Tuple, table_insert=table_insert, table_lookup=table_lookup,
                ScorePolicy=ScorePolicy)


def run_deterministic(capacities, C, batches, policy):
    stub = Stub(capacities, C)
    ns = make_ns(stub)
    for fn in functions_of(f"{REF}/dynamicemb/scored_hashtable.py", ("_bucketize_and_pad", "_deterministic_insert"),
                           cls="LinearBucketTable"):
        exec(compile(ast.Module([fn], []), "scored_hashtable.py", "exec"), ns)
    stub._bucketize_and_pad = lambda *a: ns["_bucketize_and_pad"](stub, *a)
    out = []
    for keys, tids, scores in batches:
        sc = None if scores is None else torch.from_numpy(scores.view(np.int64)).view(torch.uint64)
        idx = ns["_deterministic_insert"](stub, torch.from_numpy(keys.view(np.int64)), torch.from_numpy(tids), sc, policy)
        out.append(idx.numpy().copy())
    return out, stub.t


def flow_cases():
    rng = np.random.default_rng(2024)
    cases = {}
    # (a) two tables, small buckets, two batches: multi-wave inserts, re-inserted keys, table 1 overflowing (evictions)
    caps, C = [64, 32], 16
    b = []
    for n in (90, 70):
        keys = rng.choice(np.arange(1, 400, dtype=np.uint64), size=n, replace=False)
        tids = (rng.random(n) < 0.35).astype(np.int64)
        order = np.argsort(tids, kind="stable")
        b.append((keys[order].copy(), tids[order].copy(), rng.integers(1, 1000, n).astype(np.uint64)))
    cases["two_tables_c16"] = (caps, C, b, ScorePolicy.ASSIGN)
    # (b) one table, default bucket capacity, keys spread over 64-bit space incl. reserved ones
    keys = rng.integers(0, 2 ** 63, 500).astype(np.uint64)
    keys[:3] = [0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFE, 0xFFFFFFFFFFFFFFFC]
    keys = np.unique(keys)
    cases["one_table_c128"] = ([1024], 128, [(keys, np.zeros(keys.size, np.int64), np.arange(keys.size, dtype=np.uint64) + 5)],
                               ScorePolicy.ASSIGN)
    return cases


# ------------------------------------------------------------------------------------------------ 2. optimizer state dims
class EmbOptimType(enum.Enum):
    SGD = "sgd"
    EXACT_SGD = "exact_sgd"
    ADAM = "adam"
    EXACT_ADAGRAD = "exact_adagrad"
    EXACT_ROWWISE_ADAGRAD = "exact_row_wise_adagrad"


def optimizer_dims():
    from typing import Optional
    ns = dict(EmbOptimType=EmbOptimType, Optional=Optional, torch=torch,
              DTYPE_NUM_BYTES={torch.float32: 4, torch.float16: 2, torch.bfloat16: 2})
    for fn in functions_of(f"{REF}/dynamicemb/optimizer.py", ("get_optimizer_state_dim", "get_optimizer_ckpt_state_dim")):
        exec(compile(ast.Module([fn], []), "optimizer.py", "exec"), ns)
    rows = []
    for o in EmbOptimType:
        for dim in (7, 8, 128):
            for dt, code in ((torch.float32, 0), (torch.bfloat16, 1), (torch.float16, 2)):
                rows.append((list(EmbOptimType).index(o), dim, code, ns["get_optimizer_state_dim"](o, dim, dt),
                             ns["get_optimizer_ckpt_state_dim"](o, dim)))
    return np.array(rows, np.int64), [o.name for o in EmbOptimType]


# ------------------------------------------------------------------------------------------------ 4. planner arithmetic
def planner_arithmetic():
    """align_to_table_size / _sharded_table_bucket_layout / get_sharded_table_capacity / get_constraint_capacity
    (dynamicemb_config.py:661-760): the capacity rules the sharding planner writes into DynamicEmbTableOptions, executed
    from the reference's AST on a grid of (rows, world size, bucket capacity) and (bytes, dtype, dim, optimizer)."""
    import math
    import warnings
    from typing import Optional, Tuple

    src = f"{REF}/dynamicemb/dynamicemb_config.py"
    consts = {}
    wanted = ("DEMB_TABLE_ALIGN_SIZE", "BUCKET_ALIGNMENT", "MAX_BUCKET_CAPACITY", "DEFAULT_BUCKET_CAPACITY")
    for mod in (f"{REF}/dynamicemb/types.py", f"{REF}/dynamicemb/dynamicemb_config.py"):
        for n in ast.parse(open(mod).read()).body:
            tgt = None
            if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name):
                tgt = n.targets[0].id
            elif isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name) and n.value is not None:
                tgt = n.target.id
            if tgt in wanted:
                consts[tgt] = eval(compile(ast.Expression(n.value), mod, "eval"), dict(consts))
    ns = dict(math=math, warnings=warnings, Optional=Optional, Tuple=Tuple, torch=torch, EmbOptimType=EmbOptimType,
              BaseEmbeddingConfig=object, DTYPE_NUM_BYTES={torch.float32: 4, torch.float16: 2, torch.bfloat16: 2}, **consts)
    ns["dtype_to_bytes"] = lambda dt: ns["DTYPE_NUM_BYTES"][dt]
    for fn in functions_of(f"{REF}/dynamicemb/optimizer.py", ("get_optimizer_state_dim",)):
        exec(compile(ast.Module([fn], []), "optimizer.py", "exec"), ns)
    for fn in functions_of(src, ("align_to_table_size", "_sharded_table_bucket_layout", "get_sharded_table_capacity",
                                 "get_constraint_capacity")):
        exec(compile(ast.Module([fn], []), "dynamicemb_config.py", "exec"), ns)

    class Cfg:
        def __init__(self, n):
            self.num_embeddings = n

    rng = np.random.default_rng(9)
    align = [(int(n), int(a), ns["align_to_table_size"](int(n), int(a)))
             for n in list(rng.integers(-5, 5000, 40)) + [0, 1, 16, 17] for a in (consts["DEMB_TABLE_ALIGN_SIZE"], 128, 1024)]
    layout = []
    for n in [1, 100, 1000, 12345, 10_000_000, 999_999_937]:
        for w in (1, 2, 3, 8):
            for bc in (16, 128, 1024, consts["MAX_BUCKET_CAPACITY"]):
                nb, eff = ns["_sharded_table_bucket_layout"](Cfg(n), w, bc)
                layout.append((n, w, bc, nb, eff, ns["get_sharded_table_capacity"](Cfg(n), w, bc)))
    cap = []
    dts = [torch.float32, torch.bfloat16, torch.float16]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for mem in (1, 4096, 1 << 20, 123_456_789, 5 << 30):
            for dc, dt in enumerate(dts):
                for dim in (8, 128):
                    for oi, o in enumerate(EmbOptimType):
                        for bc in (16, 128):
                            cap.append((mem, dc, dim, oi, bc, ns["get_constraint_capacity"](mem, dt, dim, o, bc)))
    return (np.array(align, np.int64), np.array(layout, np.int64), np.array(cap, np.int64),
            np.array([consts["DEMB_TABLE_ALIGN_SIZE"], consts["BUCKET_ALIGNMENT"], consts["MAX_BUCKET_CAPACITY"]], np.int64))


# ------------------------------------------------------------------------------------------------ 5. key -> owner routing
def routing_owners():
    """`hash_key_cpu` / `assign_owner_cpu` of the reference's own CPU check of its bucketize kernel
    (test/unit_tests/test_hash_roundrobin_kuairand.py:16-45): the owner rank of a key under continuous / roundrobin /
    hash_roundrobin routing."""
    ns = dict(np=np)
    for fn in functions_of(f"{REF}/test/unit_tests/test_hash_roundrobin_kuairand.py", ("hash_key_cpu", "assign_owner_cpu")):
        exec(compile(ast.Module([fn], []), "test_hash_roundrobin_kuairand.py", "exec"), ns)
    rng = np.random.default_rng(21)
    keys = np.concatenate([rng.integers(0, 10_000_000, 3000), np.arange(0, 4000, 8),            # modulo-aliasing keys
                           rng.integers(0, 2 ** 62, 500)]).astype(np.int64)
    out = {"keys": keys}
    for W in (2, 3, 8):
        blk = (10_000_000 + W - 1) // W
        for name in ("continuous", "roundrobin", "hash_roundrobin"):
            out[f"{name}/{W}"] = ns["assign_owner_cpu"](keys, W, name, blk).astype(np.int64)
    return out


# ------------------------------------------------------------------------------------------------ 3. zipf key stream
def zipf_stream():
    ns = dict(torch=torch)
    for fn in functions_of(f"{REF}/benchmark/dataset_generator.py", ("zipf",)):
        exec(compile(ast.Module([fn], []), "dataset_generator.py", "exec"), ns)
    out = {}
    for name, (lo, hi, a, n, seed) in {"a099": (0, 5000, 0.99, 8192, 1234), "a105": (100, 1100, 1.05, 4096, 7)}.items():
        torch.manual_seed(seed)
        out[name] = (np.array([lo, hi, n, seed], np.int64), np.float64(a), ns["zipf"](lo, hi, a, n, torch.device("cpu")).numpy())
    return out


def main():
    blob = {}
    names = []
    for name, (caps, C, batches, policy) in flow_cases().items():
        idx, table = run_deterministic(caps, C, batches, policy)
        names.append(name)
        blob[f"{name}/caps"] = np.array(caps, np.int64)
        blob[f"{name}/C"] = np.array([C, int(policy), len(batches)], np.int64)
        for i, ((k, t, s), ix) in enumerate(zip(batches, idx)):
            blob[f"{name}/keys{i}"], blob[f"{name}/tids{i}"], blob[f"{name}/scores{i}"], blob[f"{name}/idx{i}"] = k, t, s, ix
        blob[f"{name}/arena"] = table.storage.copy()
        blob[f"{name}/bucket_sizes"] = table.bucket_sizes.copy()
    blob["flow_cases"] = np.array(names)
    blob["opt_rows"], optnames = optimizer_dims()
    blob["opt_names"] = np.array(optnames)
    for name, (meta, a, samples) in zipf_stream().items():
        blob[f"zipf/{name}/meta"], blob[f"zipf/{name}/alpha"], blob[f"zipf/{name}/samples"] = meta, a, samples
    blob["zipf_cases"] = np.array(["a099", "a105"])
    blob["plan_align"], blob["plan_layout"], blob["plan_capacity"], blob["plan_consts"] = planner_arithmetic()
    for k, v in routing_owners().items():
        blob[f"route/{k}"] = v
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB", names)


if __name__ == "__main__":
    main()
