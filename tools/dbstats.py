"""Top kernels of a rocprofv3 rocpd database (the default output format of this ROCm): python tools/dbstats.py x.db [N]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % (int(sys.argv[2]) if len(sys.argv) > 2 else 12)):
    print(r[0][:60].ljust(60), str(r[1]).rjust(7), ("%.3f ms" % (r[2] / 1e3)).rjust(12), ("%.1f us" % r[3]).rjust(12), "%.2f %%" % r[4])
