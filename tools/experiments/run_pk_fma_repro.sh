#!/bin/bash
# two instances of the reproducer at once on one GPU (the situation in which round 3 saw the stem weight gradient change), then one alone
cd "$(dirname "$0")"
for exe in pk_fma_repro pk_fma_repro_noslp; do
  echo "== $exe: two processes"
  ./$exe ${1:-1500} & p1=$!
  ./$exe ${1:-1500} & p2=$!
  wait $p1; wait $p2
  echo "== $exe: alone"
  ./$exe ${1:-1500}
done
