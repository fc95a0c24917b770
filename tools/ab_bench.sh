#!/bin/bash
# A/B of engine options on ONE box (boxes differ by 2-3% in clock): tools/ab_bench.sh "<flags A>" "<flags B>" ...
export PYTHONUNBUFFERED=1
for i in 1 2; do
for f in "$@"; do
python bench.py --steps 30 --warmup 5 --no-side --cpu-seconds 0 $f 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f]', d['ms_per_step'], d['value'])"
done
done
