import csv, glob, os, sys
from collections import defaultdict
src, kernel = sys.argv[1], sys.argv[2]
c = defaultdict(list)
for path in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(c):
    print(f"{k:45s} {sum(c[k])/len(c[k]):18.1f}  (n={len(c[k])})")
