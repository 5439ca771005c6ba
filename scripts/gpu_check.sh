mkdir -p gpurun_out && rm -f gpurun_out/parity_report.txt
timeout -k 10 800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "pytest exit=$rc" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out; stopping"; exit 1; fi
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; rc=$?; echo "bench exit=$rc"; tail -5 gpurun_out/bench.log
if [ $rc -ne 0 ]; then exit 1; fi
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/prof_r1; rm -rf $OUT; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof exit=$?"; ls -R $OUT | head -20
