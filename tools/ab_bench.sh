export PYTHONUNBUFFERED=1
for i in 1 2; do
for f in "" "--no-fuse-b3" "--no-fuse-b3 --no-relu-bits"; do
python bench.py --steps 30 --warmup 5 --no-side --cpu-seconds 0 $f 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['ms_per_step'], d['value'])"
done
done
