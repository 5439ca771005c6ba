# LDS-tiled generic bf16 convolution: the whole bf16 suite, then the bf16 bench lines (C3 shard, C4 frame)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gl
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/gl/tests.log 2>&1; rc=$?; echo "tests exit=$rc"; tail -3 gpurun_out/gl/tests.log
[ $rc -eq 0 ] || exit $rc
for args in "--dtype bf16" "--dtype bf16 --lr-size 540 --lr-width 960 --batch 4" "--dtype bf16 --disc simple"; do
  tag=$(echo "$args" | tr -c "A-Za-z0-9\n" "_")
  python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/gl/bench$tag.json 2> gpurun_out/gl/bench$tag.err; echo "bench [$args] exit=$?"; cut -c1-200 gpurun_out/gl/bench$tag.json
  VCG_GCONV_LDS=0 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/gl/bench${tag}_streaming.json 2> /dev/null; echo "  streaming kernel:"; cut -c80-160 gpurun_out/gl/bench${tag}_streaming.json
done
