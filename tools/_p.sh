set -e
cd /root/repo
mkdir -p gpurun_out
bash tools/prof_stats.sh r03b_infer gpurun_out/r03b_infer16_60min_kernel_stats.csv -- python3 bench.py --workload infer --cpu-seconds 0
