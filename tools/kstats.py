"""Developer tool: print a rocprofv3 kernel_stats.csv (first CSV found under the given directory) with short names."""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
for r in list(csv.DictReader(open(path)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:48]:48s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
