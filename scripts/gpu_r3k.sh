cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03k
for g in 256 128; do
rm -rf gpurun_out/prof_t
DSN_WGRAD_GROUPS=$g rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- python bench.py --train --steps 5 --warmup 2 > ${O}_prof_$g.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_t/t_results.db > ${O}_trace_$g.txt; echo groups $g; grep -h "wgrad16c\|wgrad16p" ${O}_trace_$g.txt | cut -c1-120
python scripts/rocpd_sequence.py gpurun_out/prof_t/t_results.db k_sample_gg > ${O}_seq_$g.txt
done
rm -rf gpurun_out/prof_t
