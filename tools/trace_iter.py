"""Summarise a rocprofv3 kernel trace of bench.py: per-scale wall/busy time and the per-launch durations of the last
1/4-scale iteration (of the last test_mode=True cascade).  usage: tools/trace_iter.py <kernel_trace.csv> [4|8|16]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cb = [i for i, r in enumerate(rows) if "corr_build" in r["Kernel_Name"]]
# bench.py ends with 3 test_mode=False cascades (ms_per_step_all_predictions): analyse the last test_mode=True one, the 4th from the end
ncas = len(cb) // 3
pick = ncas - 4 if ncas >= 5 else ncas - 1
s16, s8, s4 = cb[3 * pick], cb[3 * pick + 1], cb[3 * pick + 2]
end4 = cb[3 * pick + 3] if pick + 1 < ncas else len(rows)


def seg(a, b, label):
    rs = rows[a:b]
    t0, t1 = int(rs[0]["Start_Timestamp"]), int(rs[-1]["End_Timestamp"])
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    print(f"{label}: kernels={len(rs)} wall_ms={(t1 - t0) / 1e6:.2f} busy_ms={busy / 1e6:.2f} (union of kernel intervals) "
          f"sum_ms={sum(b - a for a, b in iv) / 1e6:.2f}")
    return rs


r16, r8, r4 = seg(s16, s8, "scale16"), seg(s8, s4, "scale8"), seg(s4, end4, "scale4")
which = {"16": r16, "8": r8, "4": r4}[sys.argv[2] if len(sys.argv) > 2 else "4"]
look = [i for i, r in enumerate(which) if "corr_lookup" in r["Kernel_Name"]]
it = which[look[-2]:look[-1]]
tot = 0
for r in it:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{r['Kernel_Name'][:60]:60s} grid={r['Grid_Size_X']:>8s} wg={r['Workgroup_Size_X']:>4s} lds={r['LDS_Block_Size']:>6s} dur_us={d:9.1f}")
print("iteration total us", tot)
