cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -k "loss_parity or uniform" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcf_$name -o $name -- $B > gpurun_out/pmcf_$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT TCC_MISS TCC_REQ
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/pmcf_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', ''); agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        for k in agg:
            for c, v in agg[k].items(): res[k][c] = v / n[(k, c)]
keep = {k: v for k, v in res.items() if k.startswith('k_')}
json.dump(keep, open('gpurun_out/pmc_final.json', 'w'), indent=1)
for k, v in keep.items(): print(k, {c: '%.4g' % x for c, x in v.items()})
PY
