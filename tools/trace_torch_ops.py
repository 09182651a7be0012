"""Which kernels of a bench.py kernel trace are NOT the library's (torch element-wise / copy / index kernels, runtime copies), per cascade:
count, summed duration and where they sit (inside the iteration loop of a scale or in the per-scale set-up).
usage: tools/trace_torch_ops.py <kernel_trace.csv> [cascade index, default 2]"""
import csv
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cb = [i for i, r in enumerate(rows) if "corr_build" in r["Kernel_Name"]]
assert len(cb) % 3 == 0 and len(cb) // 3 > which + 1, len(cb)
a, b = cb[3 * which], cb[3 * which + 3]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
print(f"cascade {which}: {len(seg)} kernels, {1e-6 * (t1 - t0):.2f} ms from its first corr_build to the next cascade's")
is_torch = lambda n: n.startswith("void at::") or "rocclr" in n or "at::native" in n
cnt, dur = Counter(), defaultdict(float)
in_iter = False
lookups = 0
for r in seg:
    n = r["Kernel_Name"]
    if "corr_build" in n:
        in_iter = False
    if "corr_lookup" in n:
        in_iter = True
        lookups += 1
    if is_torch(n):
        key = ("loop " if in_iter else "setup ") + n[:110]
        cnt[key] += 1
        dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot_n, tot_d = sum(cnt.values()), sum(dur.values())
print(f"non-library kernels: {tot_n} launches, {tot_d / 1e3:.3f} ms summed; iterations seen: {lookups}")
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{cnt[k]:5d} {v:9.1f} us  {k}")
