#!/bin/bash
# the two PMC summaries whose sources changed after tools/r06_pmc_all.sh ran (conv_h2.hip: the pack kernel; conv_f16.hip: the strip block)
set -eo pipefail
cs=laughter-detection-icsi_amd/csrc
PMC_SRC="$cs/conv_h2.hip $cs/lad_b3_tile.h $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_h2 "conv_h2_kernel<64" profiles/r06_conv_h2_instep_pmc.json -- python3 bench.py --steps 3 --warmup 1 --no-side --cpu-seconds 0 > gpurun_out/r06_pmc_h2.log 2>&1
rm -rf gpurun_out/r06_pmc_h2; echo h2 done
PMC_SRC="$cs/conv_f16.hip $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_blk block_f16_strip_kernel profiles/r06_block_f16_pmc.json -- python3 bench.py --workload infer --minutes 8.2 --precision fp16 --cpu-seconds 0 > gpurun_out/r06_pmc_blk.log 2>&1
rm -rf gpurun_out/r06_pmc_blk; echo block done
cp profiles/r06_conv_h2_instep_pmc.json profiles/r06_block_f16_pmc.json gpurun_out/
