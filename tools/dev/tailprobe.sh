cd /tmp && export TMPDIR=/tmp
for s in 0 1 2; do
WDX_TAIL_STOP=$s timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rf$s -- python3 $GRAFT_REPO_ROOT/tools/bench_refine.py 32768 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/rf$s -name "*kernel_stats.csv" | head -1); echo stop=$s; grep -E "refine_tail|refine_match|fingerprint_list_kernel" $f | cut -d, -f1-4 | cut -c1-150
done
