"""Mean per-launch value of every counter, per kernel, from rocprofv3 --pmc counter_collection.csv files.
usage: python tools/pmc_table.py gpurun_out/pmc_<tag>_*/p_counter_collection.csv"""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if k.startswith("at::") or "elementwise" in k or "distribution" in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
