export PYTHONUNBUFFERED=1
for i in 1 2; do
for v in -1 3 2; do
LAD_H2_VARIANT=$v python bench.py --steps 30 --warmup 5 --no-side --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[variant $v]', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])"
done
done
