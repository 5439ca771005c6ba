# the whole GPU suite, as the driver runs it (plus the 25 slowest tests)
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_report.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > gpurun_out/r2_full_tests.log 2>&1; echo "pytest exit=$?" >> gpurun_out/r2_full_tests.log; tail -45 gpurun_out/r2_full_tests.log
