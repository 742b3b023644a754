"""Per-dispatch timeline of the last bench step from a rocprofv3 --kernel-trace CSV: duration and the idle gap in front
of every dispatch.   python tools/gaps.py <dir with *kernel_trace.csv> [n_last]"""
import csv, glob, sys
fs = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=lambda f: -len(open(f).read()))
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:44]:44s} dur {(e - s) / 1e3:9.1f} us  gap_before {gap:8.1f} us  wgs {int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) // 256}")
    prev = e
