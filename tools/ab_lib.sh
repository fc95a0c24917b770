export PYTHONUNBUFFERED=1
for i in 1 2; do
for l in "" tools/libexp_h2maxilp.so tools/libexp_h2iterilp.so tools/libexp_wgmaxilp.so; do
if [ -z "$l" ]; then unset LAD_HIP_LIB; else export LAD_HIP_LIB=$PWD/$l; fi
python bench.py --steps 30 --warmup 5 --no-side --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[${l:-product}]', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])"
done
done
