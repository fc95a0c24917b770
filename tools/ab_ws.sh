#!/bin/bash
# A/B of the wave-specialised conv_h2w kernel (tools/exp_ws.sh PLAIN; LAD_H2_WS=1) against conv_h2_kernel (LAD_H2_WS=0) in the step, interleaved.
for i in 1 2; do
  for v in 0 1; do
    LAD_HIP_LIB=tools/libexp_ws_PLAIN.so LAD_H2_WS=$v python bench.py --steps 20 --warmup 5 --no-side --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[LAD_H2_WS=$v]', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['config']['final_loss'])"
  done
done
