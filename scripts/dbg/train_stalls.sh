# where do the storing training kernels wait?  separate --pmc passes (kernel-trace only) of bench.py --train; per kernel: counter sums per launch
# usage: train_stalls.sh PASS...   (sq sq2 ta tlb tcc; every pass under its own timeout: some counter sets make rocprofv3 crawl)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
C="python bench.py --train --weights default --steps 2 --warmup 1"
declare -A SETS
SETS[sq]="SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SETS[sq2]="SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"
SETS[ta]="TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"
SETS[tcp]="TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
SETS[tlb]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
SETS[tcc]="TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum"
for n in "$@"; do
  rm -rf gpurun_out/st_$n
  timeout ${PASS_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc ${SETS[$n]} --output-format csv -d gpurun_out/st_$n -o s -- $C > gpurun_out/st_$n.log 2>&1 || echo "pass $n: rc $? (timeout or failure)"
  python - $n <<'PY'
import csv, glob, collections, sys
n = sys.argv[1]
keys = ('k_field16<3>', 'k_field16ILi3', 'k_tangent16', 'k_adjoint16', 'k_t_wgrad16d')
f = glob.glob('gpurun_out/st_%s/**/*counter_collection.csv' % n, recursive=True)
if not f: print(n, 'no output'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    for k in keys:
        if k in r['Kernel_Name']:
            kk = 'k_field16<train>' if 'field16' in k else k
            acc[kk][r['Counter_Name']] += float(r['Counter_Value']); cnt[kk].add(r['Dispatch_Id'])
for k in acc:
    print(n, k, {c: '%.3g' % (v / len(cnt[k])) for c, v in acc[k].items()})
PY
  rm -rf gpurun_out/st_$n
done
