import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(r["Name"][:52].ljust(52), r["Calls"].rjust(6), ("%.3f ms" % (int(r["TotalDurationNs"]) / 1e6)).rjust(12), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12), r["Percentage"])
