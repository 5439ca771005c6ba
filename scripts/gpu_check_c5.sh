# C5 (bf16 inference) check: bench line + rocprofv3 kernel stats.  bash scripts/gpu_check_c5.sh
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout -k 10 300 python3 $R/scripts/bench_c5.py > $O/bench_c5.log 2>&1; echo "bench_c5 exit=$?"; tail -1 $O/bench_c5.log
rm -rf $O/prof_c5; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 $R/scripts/bench_c5.py --steps 5 --warmup 2 > $O/rocprof_c5.log 2>&1; echo "rocprof exit=$?"
find $O/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv
timeout -k 10 200 python3 $R/scripts/kbench_bf16.py 32 > $O/kbench_bf16.log 2>&1; cat $O/kbench_bf16.log
