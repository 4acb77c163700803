"""Launch timeline of the LAST of `calls` identical API calls in a rocprofv3 kernel trace:
start (us), duration (us), gap to the end of everything before it, queue, grid, workgroup, kernel.
Usage: python profiles/scripts/timeline.py <kernel_trace.csv> <calls>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // int(sys.argv[2])
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?')} "
          f"grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>4s} {name}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, {n} launches")
