#!/bin/bash
# (the pass directories are deleted once summarised: gpurun merges at most 64 MiB back)
# Round 6: the PMC summaries bench.py reads for roofline.traffic (each with the hashes of its kernel's sources, PMC_SRC) + the two new
# inference kernels.  The featuriser's summary stays round 5's (its sources have not changed).
set -eo pipefail
cs=laughter-detection-icsi_amd/csrc
PMC_SRC="$cs/conv_h2.hip $cs/lad_b3_tile.h $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_h2 "conv_h2_kernel<64" profiles/r06_conv_h2_instep_pmc.json -- python3 bench.py --steps 3 --warmup 1 --no-side --cpu-seconds 0 > gpurun_out/r06_pmc_h2.log 2>&1
rm -rf gpurun_out/r06_pmc_h2; echo h2 done
PMC_SRC="$cs/conv_f16.hip $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_blk block_f16_strip_kernel profiles/r06_block_f16_pmc.json -- python3 bench.py --workload infer --minutes 8.2 --precision fp16 --cpu-seconds 0 > gpurun_out/r06_pmc_blk.log 2>&1
rm -rf gpurun_out/r06_pmc_blk; echo block done
PMC_SRC="$cs/tail_f16.hip $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_tail tail_f16_kernel profiles/r06_tail_f16_pmc.json -- python3 bench.py --workload infer --minutes 8.2 --precision fp16 --cpu-seconds 0 > gpurun_out/r06_pmc_tail.log 2>&1
rm -rf gpurun_out/r06_pmc_tail; echo tail done
PMC_SRC="$cs/s2strip_f16.hip $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_s2 s2strip_f16_kernel profiles/r06_s2strip_f16_pmc.json -- python3 bench.py --workload infer --minutes 8.2 --precision fp16 --cpu-seconds 0 > gpurun_out/r06_pmc_s2.log 2>&1
rm -rf gpurun_out/r06_pmc_s2; echo s2strip done
