"""Generates tests/golden/demb_hash_golden.json by evaluating the REFERENCE's own
Python copy of the table hash (`murmur3_hash_64bits`, `uint64_to_int64`,
corelib/dynamicemb/dynamicemb/scored_hashtable.py:275-291) and its empty-digest
rule (:476-495) on a fixed key list.  The reference module itself cannot be
imported (it needs the CUDA extension), so the two pure functions are pulled
out of its AST and executed.  Run in the build container only:

    python tests/golden/gen_demb_golden.py
"""
import ast
import json
import os
import random

REF = "/root/reference/corelib/dynamicemb/dynamicemb/scored_hashtable.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demb_hash_golden.json")

src = open(REF).read()
tree = ast.parse(src)
ns = {}
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name in ("murmur3_hash_64bits", "uint64_to_int64"):
        exec(compile(ast.Module([node], []), REF, "exec"), ns)

rng = random.Random(1234)
keys = [0, 1, 2, 12, 64, 8, 15, 7, 105, 777, 100000, 2**31 - 1, 2**32, 2**63 - 1, 2**63,
        0xFFFFFFFFFFFFFFFB, 0xFFFFFFFFFFFFFFFC, 0xFFFFFFFFFFFFFFFD, 0xFFFFFFFFFFFFFFFE, 0xFFFFFFFFFFFFFFFF]
keys += [rng.getrandbits(64) for _ in range(200)]
rows = []
for k in keys:
    h = ns["murmur3_hash_64bits"](k)
    rows.append({"key": str(k), "fmix64": str(h), "digest": ((h & 0x7FFFFFFFFFFFFFFF) >> 32) & 0xFF})
empty_key = 0xFFFFFFFFFFFFFFFF
# scored_hashtable.py:482-494: empty digest computed from the int64 view of the empty key
ek = ns["uint64_to_int64"](empty_key)
empty_digest = (ns["murmur3_hash_64bits"](ek) >> 32) & 0xFF
json.dump({"source": "scored_hashtable.py:275-291,476-495", "rows": rows,
           "empty_digest": empty_digest, "empty_key_int64": ek}, open(OUT, "w"), indent=0)
print("wrote", OUT, len(rows), "rows; empty digest", empty_digest)
