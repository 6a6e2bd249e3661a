"""Summarise a rocprofv3 --kernel-trace CSV of tools/per_view_route.py: kernels per step, busy time, gaps (is the per-view route
bound by the host's issue rate, by GPU dispatch gaps, or by kernel time?).  Usage: python tools/pv_trace_summary.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# the last 40 % of the trace is steady state (warm-up + first timed variant come first)
lo = int(n * 0.1)
seg = rows[lo:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(seg, seg[1:])]
gaps_s = sorted(gaps)
print(f"kernels {len(seg)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / span:.0f} %)  mean kernel {busy / len(seg) / 1e3:.2f} us")
print(f"gap between consecutive kernels: median {gaps_s[len(gaps_s) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, p90 {gaps_s[int(0.9 * len(gaps_s))] / 1e3:.2f} us")
by = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"][:90]
    by[k][0] += 1
    by[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{c:7d} x {t / c / 1e3:7.2f} us = {t / 1e6:7.2f} ms  {k}")
