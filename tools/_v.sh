for i in 1 2 3; do
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --concurrent-clips 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('Q$q', round(d['value'],1))"
done; done
