#!/bin/bash
# PMC summaries of the step's largest kernels outside the 64-channel matrix kernels (final round-6 sources): what the "at the HBM roof" /
# "f32 MFMA at half its peak" statements of DESIGN section 8 rest on.
set -eo pipefail
cs=laughter-detection-icsi_amd/csrc
run() {  # tag, kernel substring, out, sources
  PMC_SRC="$4" tools/prof_pmc.sh "$1" "$2" "profiles/$3" -- python3 bench.py --steps 3 --warmup 1 --no-side --cpu-seconds 0 > "gpurun_out/$1.log" 2>&1
  rm -rf "gpurun_out/$1"; cp "profiles/$3" gpurun_out/; echo "$1 done"
}
run r06_pmc_bnact "bn_act_kernel<1, true>" r06_bn_act_bits_pmc.json "$cs/bn.hip $cs/lad_bn_math.h"
run r06_pmc_wgs2 "wgrad_s2_kernel<64, 32, 9" r06_wgrad_s2_64_32_pmc.json "$cs/conv_s2_bwd.hip"
run r06_pmc_dgs2 "dgrad_s2b3_kernel" r06_dgrad_s2b3_pmc.json "$cs/conv_b3.hip $cs/lad_b3_tile.h"
run r06_pmc_cs2 "conv_s2b3_kernel" r06_conv_s2b3_pmc.json "$cs/conv_b3.hip $cs/lad_b3_tile.h"
