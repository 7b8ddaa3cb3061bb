"""Busy / idle account of a rocprofv3 kernel trace between the first and last launch of a kernel whose name contains
PATTERN (last occurrence window): python tools/trace_gaps.py <kernel_trace.csv> <first-kernel> <last-kernel> [occurrence]
Prints every kernel of the window with its gap to the end of everything before it, and the union busy time."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first, last = sys.argv[2], sys.argv[3]
occ = int(sys.argv[4]) if len(sys.argv) > 4 else -1
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return n.replace("starkhip::", "").replace("void ", "")[:44]
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0 = starts[occ]
i1 = next(i for i in range(i0, len(rows)) if last in rows[i]["Kernel_Name"])
t0 = int(rows[i0]["Start_Timestamp"])
busy_end = t0
busy = 0
gaps = []
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - busy_end
    if gap > 0:
        gaps.append((gap, short(r["Kernel_Name"])))
    busy += max(0, e - max(s, busy_end))
    print("%-46s start %9.1f us dur %8.1f us gap %7.1f" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, gap / 1e3))
    busy_end = max(busy_end, e)
span = busy_end - t0
print("window %.1f us, busy %.1f us (%.1f %%), idle %.1f us in %d gaps; largest:" % (span / 1e3, busy / 1e3, 100.0 * busy / span, (span - busy) / 1e3, len(gaps)))
for g, n in sorted(gaps, reverse=True)[:8]:
    print("   %.1f us before %s" % (g / 1e3, n))
