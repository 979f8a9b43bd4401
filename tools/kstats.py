"""Print selected rows of a rocprofv3 kernel_stats.csv: python tools/kstats.py FILE [substr ...]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:]
for r in rows:
    n = r["Name"]
    short = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
    if not keys or any(k in n for k in keys):
        print("%-44s calls %5s avg %9.1f us  total %9.2f ms" % (short[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
