set -eo pipefail
cs=laughter-detection-icsi_amd/csrc
PMC_SRC="$cs/conv_f16.hip $cs/lad_device.h" tools/prof_pmc.sh r05_pmc_blk block_f16_strip_kernel profiles/r05_block_f16_pmc.json -- python3 bench.py --workload infer --minutes 8.2 --precision fp16 --cpu-seconds 0 > gpurun_out/r05_pmc_blk.log 2>&1
head -3 $(find gpurun_out/r05_pmc_blk/pass0 -name '*counter_collection.csv' | head -n 1) | cut -c1-600
rm -rf gpurun_out/r05_pmc_blk; cp profiles/r05_block_f16_pmc.json gpurun_out/
