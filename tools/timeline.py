"""diagnostic only: kernel timeline of the LAST frame in a rocprofv3 --kernel-trace rocpd database, per HIP stream, consecutive
dispatches of one kernel collapsed:  python tools/timeline.py x.db [min_gap_us]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, stream, start, end from kernels order by start").fetchall()
t0 = max(r[2] for r in rows if "job_scan" in r[0])
rows = [r for r in rows if r[2] >= t0]
streams = {}
for name, stream, s, e in rows:
    streams.setdefault(stream, []).append((name.split("(")[0].replace("rhccq::", "").replace("void ", "")[:34], (s - t0) / 1e3, (e - t0) / 1e3))
for stream, ev in streams.items():
    print("==", stream)
    i = 0
    while i < len(ev):
        j = i
        busy = 0.0
        while j < len(ev) and ev[j][0] == ev[i][0]:
            busy += ev[j][2] - ev[j][1]
            j += 1
        print(f"  {ev[i][1] / 1e3:9.3f} .. {ev[j - 1][2] / 1e3:9.3f} ms  {ev[i][0]:34s} x{j - i:<6d} busy {busy / 1e3:8.3f} ms")
        i = j
