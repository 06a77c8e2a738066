"""Reduces two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --graph 0` to HBM bytes per training
step and writes profiles/rNN_pmc_traffic.json (with the fingerprint of the kernel sources it was taken with), which bench.py reports as roofline.traffic.

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): both counters are in KiB; FETCH_SIZE
reports exactly half the bytes of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is exact for
16-B-per-lane stores.  Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [workload tag]
"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_fingerprint

def last_step(d):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "k_pack_input" in r["Kernel_Name"]]
    seg = rows[idx[-1]:]
    agg = collections.defaultdict(float)
    for r in seg:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k] += float(r["Counter_Value"]) * 1024.0
    return agg, len(seg)

fetch, n1 = last_step(sys.argv[1])
write, n2 = last_step(sys.argv[2])
rd = 2.0 * sum(fetch.values())
wr = sum(write.values())
per_kernel = {k: {"read_GB": round(2.0 * fetch.get(k, 0) / 1e9, 3), "write_GB": round(write.get(k, 0) / 1e9, 3)}
              for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))[:14]}
out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "resnet S=7 batch 64", "dispatches_per_step": n1,
       "source_fingerprint": source_fingerprint(),
       "read_bytes_per_step": rd, "write_bytes_per_step": wr, "hbm_bytes_per_step": rd + wr,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --graph 0; KiB counters; "
                 "FETCH_SIZE x2 (gfx950 wide-read correction); last training step of the run",
       "per_kernel": per_kernel}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("read %.2f GB + write %.2f GB = %.2f GB per step" % (rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
