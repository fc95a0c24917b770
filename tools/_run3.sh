set -e
echo "=== wgrad tests (variant 1 default)"
timeout -k 10 400 python -m pytest tests/test_resnet_gpu.py -x -q -m gpu -k "wgrad or virtual or golden or oracle or fused" 2>&1 | tail -5
echo "=== microbench wgrad"
timeout -k 10 300 python tools/bench_conv.py wgradb3 --iters 20 --variant 0 1 --rounds 3
echo "=== in-step wgrad variant 0 / 1"
for v in 0 1 0 1; do
LAD_WGRAD_B3_VARIANT=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-side --cpu-seconds 0 --blocks 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[wgrad variant $v]', d['ms_per_step'], d['value'], d['timed_blocks'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
