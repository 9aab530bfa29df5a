cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03m
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_EA_W[A-Z_0-9]*\|TCP_[A-Z_0-9]*STALL[A-Z_0-9]*\|TCC_[A-Z_]*STALL[A-Z_0-9]*\|TA_[A-Z_0-9]*BUSY[A-Z_0-9]*" | sort -u > ${O}_avail.txt
wc -l ${O}_avail.txt
rm -rf gpurun_out/pmcm
runt() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcm/$name -o $name -- python bench.py --train --steps 3 --warmup 2 > gpurun_out/pmcm_$name.log 2>&1; }
runt a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
runt b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR
runt c SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS
runt d TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum TCC_EA_WR_UNCACHED_32B_sum
runt e TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
runt f SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SALU
runt g GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU
python - <<'PY'
import csv, glob, collections
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmcm/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('k_field16<3>','k_field16<(int)3>','k_tangent16','k_adjoint16','k_t_wgrad16c'):
            if key in k or (key=='k_field16<3>' and 'k_field16' in k):
                res[key if 'field16' not in key else 'k_field16<train>'][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in res.items():
    print(k)
    for c,x in sorted(v.items()):
        print('   %-40s %16.0f  (n=%d)'%(c, sum(x)/len(x), len(x)))
PY
rm -rf gpurun_out/pmcm
