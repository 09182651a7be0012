"""Timeline of ONE PPMStereo.forward_batch_test(host video) call (SURVEY 8d's wall-time definition) from a rocprofv3 kernel trace of tools/whole_call_probe.py:
which part of the call's GPU span each phase takes (feature encoder, context encoder, SST block, the three scales of the loop), how much of the span the GPU is
busy, and the largest idle gaps with the kernels on either side.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o trace -- /usr/bin/python3 <repo>/tools/whole_call_probe.py
    python tools/whole_call_timeline.py <dir>/.../trace_kernel_trace.csv > profiles/rNN_whole_call_timeline.txt
The probe's first timed block runs whole calls back to back; the LAST complete call before its phase-by-phase section is analysed (calls are delimited by
the feature encoder's first kernel, nchw_to_sp / the space-to-depth conv of conv1)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
cb = [i for i, r in enumerate(rows) if "corr_build" in name(r)]
assert len(cb) >= 3 * 7, "expected the probe's 7 whole calls (3 correlation builds each)"
# whole calls 0..6 are the first 7 triples of corr_build launches; analyse call 6 (the last of the 5 timed ones)
call = 6
lo_cb, hi_cb = cb[3 * call], cb[3 * call + 2]
prev_end = cb[3 * (call - 1) + 2]
# the call starts at the first kernel after the previous call's last kernel: search the end of the previous cascade = the last kernel before a gap > 150 us
start = prev_end
for i in range(prev_end, lo_cb):
    if int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) > 150_000:
        start = i + 1
nxt = cb[3 * (call + 1)] if len(cb) > 3 * (call + 1) else len(rows)
end = hi_cb
for i in range(hi_cb, nxt - 1):
    if int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) > 150_000:
        end = i
        break
seg = rows[start:end + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)


def busy(rs):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > ce:
            tot += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    return tot + ce - cs


print(f"one forward_batch_test call (T=5, 320x512, iters=10): {len(seg)} kernels, GPU span {(t1 - t0) / 1e6:.2f} ms, busy {busy(seg) / 1e6:.2f} ms "
      f"(union of kernel intervals), sum of kernel durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6:.2f} ms")
# phases by position: everything before the first corr_build = encoders + SST + glue; then the three scales
i16, i8, i4 = (cb[3 * call + k] - start for k in range(3))
for label, a, b in (("before the loop (H2D copy kernels, fnet, cnet, SST, feature pyramids)", 0, i16), ("1/16 scale", i16, i8), ("1/8 scale", i8, i4), ("1/4 scale + output", i4, len(seg))):
    rs = seg[a:b]
    s0, s1 = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
    print(f"  {label:75s} {len(rs):4d} kernels  span {(s1 - s0) / 1e6:6.2f} ms  busy {busy(rs) / 1e6:6.2f} ms")
pre = seg[:i16]
groups = {}
for r in pre:
    k = name(r).replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    g = groups.setdefault(k, [0, 0])
    g[0] += 1
    g[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("  kernels before the loop, by name (launches, total us):")
for k, (n, d) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"      {k[:70]:70s} {n:4d}  {d / 1e3:8.1f}")
gaps = []
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r)) for r in seg)
ce, cn = iv[0][1], iv[0][2]
for a, b, n in iv[1:]:
    if a > ce:
        gaps.append((a - ce, cn, n, (ce - t0) / 1e6))
    if b > ce:
        ce, cn = b, n
print(f"  idle inside the span: {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)} gaps; the largest:")
for g, a, b, at in sorted(gaps, reverse=True)[:12]:
    print(f"      {g / 1e3:7.1f} us at {at:6.2f} ms  after {a[:50]:50s} before {b[:50]}")
