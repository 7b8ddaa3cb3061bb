"""Kernel timeline of the last resident batch-verification call in a rocprofv3 kernel trace:
python tools/resident_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return n.replace("starkhip::", "").replace("batch::", "").replace("void ", "")[:34]
idx = [i for i, r in enumerate(rows) if "vio_kernel" in r["Kernel_Name"]]
i = idx[-2]
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[i:idx[-1]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-36s start %8.1f us  end %8.1f us  dur %7.1f us  grid %s" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r["Grid_Size_X"]))
