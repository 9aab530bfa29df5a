"""Dump the GPU timeline (kernels + memory copies, start-ordered, with idle gaps) of the last N ms of a rocprofv3
rocpd SQLite result.  usage: rocpd_timeline.py results.db [window_ms=80]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
ev = [(s, e, n) for n, s, e in db.execute(
    f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id")]
mc = [t for t in tabs if t.startswith('rocpd_memory_copy')]
if mc:
    cols = [r[1] for r in db.execute(f"pragma table_info({mc[0]})")]
    sz = "size" if "size" in cols else "0"
    ev += [(s, e, f"<memcpy {b} B>") for s, e, b in db.execute(f"select start, end, {sz} from {mc[0]}")]
ev.sort()
t_end = ev[-1][1]
ev = [x for x in ev if x[0] >= t_end - win * 1e6]
t0, prev, busy = ev[0][0], ev[0][0], 0
for s, e, n in ev:
    gap = (s - prev) / 1e3
    busy += (e - max(s, prev)) if e > prev else 0
    print(f"{(s - t0) / 1e6:9.3f} ms  +{gap:8.1f} us idle  {(e - s) / 1e3:9.1f} us  {n[:90]}")
    prev = max(prev, e)
print(f"window {(prev - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
