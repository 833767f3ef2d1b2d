"""Turns the PMC dump of tools/pmc_run.sh into profiles/<round>_pmc_traffic.json: HBM traffic per launch of the
bandwidth kernels = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH_SIZE counts 64 B per 128-B request on gfx950,
MI355X_MICROARCH.md HBM section), cross-checked with the TCC_EA0 request counts (RDREQ*128 B, WRREQ*64 B).
Usage: python tools/pmc_traffic.py <pmc dump txt> <out json>"""
import json
import re
import sys

src, out = sys.argv[1], sys.argv[2]
vals = {}
for line in open(src):
    m = re.match(r"(.{60}) (\S+)\s+n=\s*\d+ avg=\s*([\d.]+)", line)
    if not m:
        continue
    name = re.sub(r"^void ", "", m.group(1).strip())
    name = re.sub(r"^mi355::", "", name).split("(")[0].split("<")[0]
    vals.setdefault(name, {})[m.group(2)] = float(m.group(3))
kern = {}
for name, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
             "traffic_bytes": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024}
        if "TCC_EA0_RDREQ_sum" in v:
            e["TCC_EA0_RDREQ"] = v["TCC_EA0_RDREQ_sum"]
            e["TCC_EA0_WRREQ"] = v.get("TCC_EA0_WRREQ_sum")
            e["traffic_bytes_from_requests"] = v["TCC_EA0_RDREQ_sum"] * 128 + (v.get("TCC_EA0_WRREQ_sum") or 0) * 64
        kern[name] = e
json.dump({"source": src + " (tools/pmc_run.sh: rocprofv3 --pmc passes over bench.py --steps 8)",
           "formula": "traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); cross-check: TCC_EA0_RDREQ_sum*128 B + TCC_EA0_WRREQ_sum*64 B",
           "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps({k: round(v["traffic_bytes"] / 1e6, 1) for k, v in kern.items()}))
