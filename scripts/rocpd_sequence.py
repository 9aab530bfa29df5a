"""List the kernel dispatches of the LAST step of a rocprofv3 rocpd result in launch order: name, duration (ms), gap to the previous
kernel's end (us).  usage: rocpd_sequence.py results.db [first-kernel-of-a-step substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
mark = sys.argv[2] if len(sys.argv) > 2 else "k_sample_gg"
starts = [i for i, r in enumerate(rows) if mark in r[0]]
lo = starts[-1] if starts else 0
prev = None
for name, s, e in rows[lo:]:
    short = name.split("(")[0][:70]
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{short:70s} {(e - s) / 1e6:8.3f} ms   gap {gap:7.1f} us")
    prev = e
print(f"step: {(rows[-1][2] - rows[lo][1]) / 1e6:.3f} ms from first start to last end")
