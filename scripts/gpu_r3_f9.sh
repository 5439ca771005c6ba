cd $GRAFT_REPO_ROOT
python scripts/micro/f9_stamps.py 8 512 512 30 && python scripts/micro/f9_stamps.py 4 1080 1920 10
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "final_conv or generator" 2>&1 | tail -3
python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline | cut -c1-400
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline | cut -c1-200
