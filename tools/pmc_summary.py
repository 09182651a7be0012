"""Summarise a rocprofv3 --pmc csv (counter_collection.csv): the LAST dispatch of each kernel whose name contains <filter>,
every counter (and, with --all, every dispatch)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
disp = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if filt and filt not in k:
        continue
    disp.setdefault((int(r["Dispatch_Id"]), k[:50], r.get("Grid_Size", "")), {})[r["Counter_Name"]] = float(r["Counter_Value"])
items = list(disp.items())
for (did, k, grid), d in (items if "--all" in sys.argv else items[-3:]):
    print(did, k, "grid", grid, " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
