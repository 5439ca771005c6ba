# usage: bash scripts/pmc_step.sh <tag> <steps> <bench args...>
# HBM traffic of a whole bench.py step: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md
# "rocprofv3 PMC slots": FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), kernels launched eagerly (--no-graph) so that every
# dispatch is a record.  <steps> = timed steps per pass (plus 1 warm-up step); summarise with scripts/pmc_step_summary.py.
TAG=$1; STEPS=$2; shift 2
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcstep_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p1 -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-graph "$@" > $OUT/p1.log 2>&1; echo "pmcstep $TAG fetch exit=$?"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p2 -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-graph "$@" > $OUT/p2.log 2>&1; echo "pmcstep $TAG write exit=$?"
