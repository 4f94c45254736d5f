"""Re-summarise the counter CSVs of tools/pmc_kernels.sh: per kernel and counter the number of launches,
the mean per launch and the sum.  usage: tools/pmc_summary.py gpurun_out/<tag> <pass> [<pass> ...]"""
import collections
import csv
import glob
import os
import sys

O = sys.argv[1]
for d in sys.argv[2:]:
    f = sorted(glob.glob(f"{O}/{d}/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if k.startswith("at::") or "elementwise" in k or k.startswith("__amd_rocclr"):
            continue
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{d:6s} {k[:40]:40s} {c:24s} launches {len(v):5d} mean {sum(v)/len(v):16.1f} sum {sum(v):18.1f}")
