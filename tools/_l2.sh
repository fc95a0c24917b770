set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_resnet_gpu.py -x -q -m gpu -k "sharing_the_second or streaming_sliding or fp16_inference or sliding_window" -s > gpurun_out/l2_tests.log 2>&1 || { tail -40 gpurun_out/l2_tests.log; exit 1; }
tail -8 gpurun_out/l2_tests.log
timeout -k 10 600 python bench.py --workload infer --cpu-seconds 0 > gpurun_out/l2_bench.log 2>&1 || true
tail -5 gpurun_out/l2_bench.log
