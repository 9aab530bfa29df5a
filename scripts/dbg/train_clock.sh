# shader clock of the three storing training kernels per library variant: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for so in dual-space-nerf_amd/variants/*.so; do
  n=$(basename $so .so); D=gpurun_out/clk_$n; rm -rf $D
  DSNERF_LIB=$PWD/$so rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $D -o c -- python bench.py --train --weights default --steps 4 --warmup 2 > /dev/null 2>&1
  echo "== $n"
  python - $D <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cf = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
kf = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
dur = {}
for r in csv.DictReader(open(kf[0])):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cf[0])):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    name, ns = dur.get(r['Dispatch_Id'], (r['Kernel_Name'], 0))
    for key in ('k_field16ILi3', 'k_tangent16', 'k_adjoint16', 'k_t_wgrad16d'):
        if key in name:
            a = acc[key]; a[0] += float(r['Counter_Value']); a[1] += ns; a[2] += 1
for k, (cyc, ns, n) in acc.items():
    print('  %-16s launches %3d  avg %.3f ms  cycles/XCD %.3e  clock %.2f GHz' % (k, n, ns / n * 1e-6, cyc / 8 / n, cyc / 8 / ns))
PY
  rm -rf $D
done
