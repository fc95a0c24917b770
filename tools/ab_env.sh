#!/bin/bash
# A/B of an environment knob on ONE box: tools/ab_env.sh NAME "v1 v2 ..." [bench flags]
export PYTHONUNBUFFERED=1
name=$1; vals=$2; shift 2
for i in 1 2; do
for v in $vals; do
env $name=$v python bench.py --steps 30 --warmup 5 --no-side --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$name=$v]', d['ms_per_step'], d['value'])"
done
done
