"""Print the last N kernel dispatches of a rocprofv3 --kernel-trace csv with their durations and the gaps between them
(tools/session.sh traceT): where the time of one apply goes launch by launch."""
import csv, glob, sys

def main():
    d, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-last:]
    prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][:70]
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print(f"{(e - s) / 1e3:10.1f} us  gap {gap:8.1f} us  grid {r.get('Grid_Size_X', '?'):>9}  {name}")
        prev = e

if __name__ == "__main__":
    main()
