"""Summarise gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if "qcdft" in r["Kernel_Name"]]
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(ours) - 40
last = ours[n0:n0 + 14]
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{name:30s} dur {(e - s) / 1e3:8.1f} us   gap before {gap:7.1f} us")
    prev_end = e
