#!/usr/bin/env python3
"""HBM bytes per launch and kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the same command.
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced streaming read (MI355X_MICROARCH.md, section HBM).
usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> "<note>" """
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(root, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
            name = re.sub(r"\(.*$", "", name)
            acc[name][0] += float(r["Counter_Value"])
            acc[name][1] += 1
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("at::") or "rocclr" in k:
        continue
    fs, fn = fetch.get(k, [0.0, 0])
    ws, wn = write.get(k, [0.0, 0])
    n = max(fn, wn)
    if n == 0:
        continue
    fkb, wkb = (fs / fn if fn else 0.0), (ws / wn if wn else 0.0)
    out["kernels"][k] = {"launches": n, "fetch_KB_raw": round(fkb, 1), "write_KB": round(wkb, 1), "hbm_bytes_per_launch": int((2 * fkb + wkb) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(out['kernels'])} kernels -> {sys.argv[3]}")
