# rocprofv3 kernel stats of the mixed-precision step (bf16 trunk).  bash scripts/gpu_prof_mixed.sh
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
rm -rf $O/prof_mixed; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mixed -o mixed -- python3 $R/bench.py --trunk-dtype ${1:-bf16} --steps 3 --warmup 1 > $O/rocprof_mixed.log 2>&1; echo "rocprof exit=$?"
find $O/prof_mixed -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/mixed_kernel_stats.csv
