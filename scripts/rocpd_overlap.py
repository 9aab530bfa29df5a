"""Which kernels of a pipelined run hide behind the matrix-bound field kernels, and which are exposed?

usage: rocpd_overlap.py results.db [frames_to_skip=3]     (rocprofv3 --kernel-trace of `bench.py --pipeline 2 ...`)

For every kernel name: launches, total duration, and the part of that duration during which NO field kernel (k_screen16 /
k_field16<forward|reverse>) of any stream was running = its exposed time.  Sums are over the steady-state window (the first
`frames_to_skip` forward launches are skipped) and divided by the number of frames in it (= forward launches)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
ev = sorted((s, e, n) for n, s, e in db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id"))


def is_field(n):
    return ("k_screen16" in n) or ("k_field16" in n)


fwd = [x for x in ev if "k_field16ILi1E" in x[2]]
if len(fwd) <= skip + 1:
    sys.exit("not enough frames in the trace")
t_lo, t_hi = fwd[skip][0], fwd[-1][0]           # window: from the (skip+1)-th forward launch to the last one (whole frames)
frames = len(fwd) - 1 - skip
win = [x for x in ev if x[0] >= t_lo and x[0] < t_hi]
# union of the field kernels' intervals
iv = sorted((s, e) for s, e, n in win if is_field(n))
merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])


def covered(s, e):
    c = 0
    for a, b in merged:
        if b <= s:
            continue
        if a >= e:
            break
        c += min(e, b) - max(s, a)
    return c


stat = {}
for s, e, n in win:
    key = n.split("(")[0][:70]
    d = stat.setdefault(key, [0, 0, 0])
    d[0] += 1
    d[1] += e - s
    d[2] += (e - s) - (0 if is_field(n) else covered(s, e))
field_union = sum(b - a for a, b in merged)
allv = sorted((s, e) for s, e, n in win)
um = []
for s, e in allv:
    if um and s <= um[-1][1]:
        um[-1][1] = max(um[-1][1], e)
    else:
        um.append([s, e])
busy = sum(b - a for a, b in um)
print(f"steady-state window: {frames} frames, {(t_hi - t_lo) / 1e6 / frames:.3f} ms per frame; GPU busy (union of all kernels) "
      f"{busy / 1e6 / frames:.3f} ms, field kernels (union) {field_union / 1e6 / frames:.3f} ms, idle {((t_hi - t_lo) - busy) / 1e6 / frames:.3f} ms per frame")
print(f"{'kernel':70s} {'per frame':>9s} {'ms/frame':>9s} {'exposed ms/frame':>17s}")
tot_exposed = 0.0
for k, (c, d, x) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    f = is_field(k)
    if not f:
        tot_exposed += x
    print(f"{k:70s} {c / frames:9.2f} {d / 1e6 / frames:9.3f} {'(field)' if f else f'{x / 1e6 / frames:.3f}':>17s}")
print(f"non-field kernel time NOT under a field kernel: {tot_exposed / 1e6 / frames:.3f} ms per frame")
