# usage: bash scripts/gpu_prof_bench.sh <tag> [bench args...]: rocprofv3 --kernel-trace --stats of bench.py -> gpurun_out/prof_<tag>/
TAG=$1; shift
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/rocprof.log; echo "rocprof exit=$?"
find $OUT -name "*kernel_stats.csv" | head -3
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print("%-64s calls=%5s avg_us=%9.1f pct=%5.2f" % (r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:64], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
print("total ms", tot/1e6)
PY
