#!/usr/bin/env python3
"""Mean FETCH_SIZE (KiB as rocprofv3 reports it, x 1024 x 2: the gfx950 correction of tools/summarize_prof.py) and duration per kernel
from the counter-collection CSVs under the given directories.  usage: fetch_summary.py label=dir ..."""
import csv, glob, os, sys, collections
tab = collections.OrderedDict()
for spec in sys.argv[1:]:
    label, d = spec.split("=", 1)
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE": continue
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            a = acc[k]; a[0] += float(r["Counter_Value"]) * 2048 / 1e6; a[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; a[2] += 1
    tab[label] = acc
keys = sorted({k for a in tab.values() for k in a}, key=lambda k: -max(a[k][0] for a in tab.values() if k in a))
print(f"{'kernel':44s}" + "".join(f" | {l:>22s}" for l in tab))
print(f"{'(fabric reads MB per launch, us per launch)':44s}")
for k in keys[:14]:
    print(f"{k[:44]:44s}" + "".join(f" | {a[k][0]/max(a[k][2],1):9.1f} MB {a[k][1]/max(a[k][2],1):7.1f} us" if k in a else " | " + " " * 22 for a in tab.values()))
print(f"{'sum over the pass (MB)':44s}" + "".join(f" | {sum(v[0] for v in a.values())/2:12.0f} per pass   " for a in tab.values()))
