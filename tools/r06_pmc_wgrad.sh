#!/bin/bash
# PMC summaries of the two wgrad_h2 forms in the step, final round-6 sources (the kernel VERDICT r5 named furthest below its roof).
set -eo pipefail
cs=laughter-detection-icsi_amd/csrc
PMC_SRC="$cs/wgrad_mfma.hip $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_wg1 "wgrad_h2_kernel<true" profiles/r06_wgrad_h2_bnbwd_instep_pmc.json -- python3 bench.py --steps 3 --warmup 1 --no-side --cpu-seconds 0 > gpurun_out/r06_pmc_wg1.log 2>&1
rm -rf gpurun_out/r06_pmc_wg1; echo fused done
PMC_SRC="$cs/wgrad_mfma.hip $cs/lad_device.h" tools/prof_pmc.sh r06_pmc_wg2 "wgrad_h2_kernel<false" profiles/r06_wgrad_h2_instep_pmc.json -- python3 bench.py --steps 3 --warmup 1 --no-side --cpu-seconds 0 > gpurun_out/r06_pmc_wg2.log 2>&1
rm -rf gpurun_out/r06_pmc_wg2; echo plain done
cp profiles/r06_wgrad_h2_bnbwd_instep_pmc.json profiles/r06_wgrad_h2_instep_pmc.json gpurun_out/
