#!/bin/bash
# the reproducer next to a DIFFERENT busy process: the training backward loop (tools/diag_determinism.py --inproc), whose kernels
# fill whole CUs (LDS, registers) -- two copies of the small reproducer can share CUs and need not be time-sliced inside a kernel
cd "$(dirname "$0")/../.."
python tools/diag_determinism.py --inproc 100000 --batch 64 > /dev/null 2>&1 &
peer=$!
sleep 14
for exe in pk_fma_repro pk_fma_repro_noslp pk_fma_repro; do
  echo "== $exe next to the training loop"
  tools/experiments/$exe ${1:-4000}
done
kill $peer
wait $peer 2>/dev/null
echo "== pk_fma_repro alone"
tools/experiments/pk_fma_repro ${1:-4000}
