"""diagnostic only: per-stream phases of every frame in a rocprofv3 --kernel-trace rocpd database (k-means++ chain, mini-batch
steps, assignment) and the durations of the step kernels over time:  python tools/dbphases.py x.db [frame ...]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, stream, start, end from kernels order by start").fetchall()
scans = [r[2] for r in rows if "job_scan" in r[0]]
frames = [int(a) for a in sys.argv[2:]] or [-2, -1]
for fi in frames:
    t0 = scans[fi]
    t1 = scans[fi + 1] if fi not in (-1, len(scans) - 1) else 1e30
    fr = [r for r in rows if t0 <= r[2] < t1]
    by = {}
    for n, s, a, b in fr:
        by.setdefault(s, []).append((n.split("(")[0].replace("rhccq::", "").replace("void ", "")[:28], (a - t0) / 1e3, (b - t0) / 1e3))
    print(f"=== frame {fi}: {(max(r[3] for r in fr) - t0) / 1e6:.2f} ms")
    for s, ev in sorted(by.items()):
        ini = [e for e in ev if e[0].startswith("mbk_init")]
        st = [e for e in ev if e[0].startswith(("mbk_update", "mbk_pipe", "mbk_batch_estep", "mbk_fold", "mbk_fix"))]
        if not st:
            continue
        print(f"  {s:10s} chain {[(round(a / 1e3, 2), round(b / 1e3, 2)) for _, a, b in ini]} steps {st[0][1] / 1e3:.2f}..{st[-1][2] / 1e3:.2f} ms ({len(st)} launches)")
        if len(st) > 400:
            for kn in ("mbk_batch_estep", "mbk_fold", "mbk_update", "mbk_fix", "mbk_pipe"):
                sel = [(a, b - a) for n, a, b in ev if n.startswith(kn)]
                for i in range(0, len(sel), 250):
                    ch = sel[i:i + 250]
                    if len(ch) > 20:
                        print(f"       {kn:16s} {i:5d}..: t={ch[0][0] / 1e3:7.2f} ms  mean {sum(d for _, d in ch) / len(ch):6.2f} us  max {max(d for _, d in ch):7.2f}"
                              f"  period {(ch[-1][0] - ch[0][0]) / (len(ch) - 1):6.2f} us")
