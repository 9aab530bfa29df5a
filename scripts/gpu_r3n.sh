cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03n
rm -rf gpurun_out/pmcn
B="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --pipeline 1"
runt() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcn/$name -o $name -- $B > gpurun_out/pmcn_$name.log 2>&1; }
runt a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
runt c SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS
runt e TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
runt g GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU
python - <<'PY'
import csv, glob, collections
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmcn/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('k_field16','k_screen16'):
            if key in k:
                res[k.split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in res.items():
    print(k)
    for c,x in sorted(v.items()):
        print('   %-40s %16.0f  (n=%d, max %.0f)'%(c, sum(x)/len(x), len(x), max(x)))
PY
rm -rf gpurun_out/pmcn
