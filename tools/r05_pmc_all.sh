#!/bin/bash
# (the pass directories are deleted once summarised: gpurun merges at most 64 MiB back)
# Round 5: the PMC summaries bench.py reads for roofline.traffic, each with the hashes of its kernel's sources (PMC_SRC).
set -eo pipefail
cs=laughter-detection-icsi_amd/csrc
PMC_SRC="$cs/conv_h2.hip $cs/lad_b3_tile.h $cs/lad_device.h" tools/prof_pmc.sh r05_pmc_h2 "conv_h2_kernel<64" profiles/r05_conv_h2_instep_pmc.json -- python3 bench.py --steps 3 --warmup 1 --no-side --cpu-seconds 0 > gpurun_out/r05_pmc_h2.log 2>&1
rm -rf gpurun_out/r05_pmc_h2; echo h2 done
# (the 1024-clip launches only: the workload also runs the kernel over the 60-minute channel, on the same persistent grid)
LAD_BENCH_FBANK_CHANNEL=0 PMC_SRC="$cs/fbank16.hip $cs/lad_fbank16.h" tools/prof_pmc.sh r05_pmc_fb fbank16_kernel profiles/r05_fbank_pmc.json -- python3 bench.py --workload fbank --steps 20 --warmup 5 > gpurun_out/r05_pmc_fb.log 2>&1
rm -rf gpurun_out/r05_pmc_fb; echo fbank done
PMC_SRC="$cs/conv_f16.hip $cs/lad_device.h" tools/prof_pmc.sh r05_pmc_blk block_f16_strip_kernel profiles/r05_block_f16_pmc.json -- python3 bench.py --workload infer --minutes 8.2 --precision fp16 --cpu-seconds 0 > gpurun_out/r05_pmc_blk.log 2>&1
rm -rf gpurun_out/r05_pmc_blk; echo block done
