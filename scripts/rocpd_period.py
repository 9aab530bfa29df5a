"""Per step of a rocprofv3 rocpd result: period between consecutive occurrences of a marker kernel, the kernel time inside it and the
idle time (gaps between one kernel's end and the next one's start), with the ten largest gaps of the last step.
usage: rocpd_period.py results.db [marker substring = k_sample_gg_bounds]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
mark = sys.argv[2] if len(sys.argv) > 2 else "k_sample_gg_bounds"
idx = [i for i, r in enumerate(rows) if mark in r[0]]
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b + 1]
    busy = sum(e - s for _, s, e in seg[:-1])
    period = seg[-1][1] - seg[0][1]
    print(f"period {period / 1e6:8.3f} ms   kernels {busy / 1e6:8.3f} ms   idle {(period - busy) / 1e6:7.3f} ms   launches {len(seg) - 1}")
if len(idx) >= 2:
    seg = rows[idx[-2]:idx[-1] + 1]
    gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, seg[i][0].split("(")[0][:50], seg[i + 1][0].split("(")[0][:50]) for i in range(len(seg) - 1))[::-1]
    for g, x, y in gaps[:10]:
        print(f"  gap {g:8.1f} us after {x} before {y}")
