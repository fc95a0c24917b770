set -e
cd /root/repo
mkdir -p gpurun_out
bash tools/prof_pmc.sh r03_pmc_f16q conv_f16_s1q_kernel gpurun_out/r03_conv_f16q_pmc.json -- python3 bench.py --workload infer --minutes 8.2 --precision fp16 --cpu-seconds 0 > /dev/null
