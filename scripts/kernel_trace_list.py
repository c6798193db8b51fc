"""List the dispatches of one kernel from a rocprofv3 --kernel-trace CSV in launch order: start offset and duration."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = len(rows) // 2
rows = rows[half:]                      # the second call (warm)
t0 = int(rows[0]["Start_Timestamp"])
print("n launches", len(rows), "total span ms", (int(rows[-1]["End_Timestamp"]) - t0) / 1e6, "sum of durations ms", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6)
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{i:3d} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?')}")
