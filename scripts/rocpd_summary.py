"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into a per-kernel table (text)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6, "
     f"max(d.end-d.start)/1e6, max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
rows = list(db.execute(q))
tot = sum(r[2] for r in rows)
print(f"{'kernel':58s} {'calls':>5s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>6s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{r[0][:58]:58s} {r[1]:5d} {r[2]:10.3f} {r[3]:9.3f} {r[4]:9.3f} {r[5]:9.3f} {100*r[2]/tot:6.1f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:6d}")
