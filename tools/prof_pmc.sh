#!/bin/bash
# PMC passes of one command under rocprofv3, one counter group per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
# do not fit one pass; --pmc is never combined with a trace domain), then tools/pmc_summary.py over the passes.
#
#   [PMC_SRC="csrc file ..."] tools/prof_pmc.sh <tag> <kernel substring> <out.json> -- python3 <script> [args...]
#   PMC_GRID_MAX: only dispatches of at most this many work-items enter the summary
#   PMC_SRC: sources of the measured kernel (relative to the repository root); their hashes go into the summary (pmc_summary.py --src)
#
# Output: gpurun_out/<tag>/pass<k>/ (scratch) and <out.json> (the summary that gets committed under profiles/).
set -eo pipefail
tag=$1; kernel=$2; out=$3; shift 3
[ "$1" == "--" ] && shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# the command runs from /tmp: make its repo-relative file arguments absolute
args=()
for a in "$@"; do
  if [ -e "$root/$a" ] && [[ "$a" != /* ]]; then args+=("$root/$a"); else args+=("$a"); fi
done
[[ "$out" != /* ]] && out="$root/$out"
cd /tmp && export TMPDIR=/tmp
groups=(
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
csvs=()
k=0
for g in "${groups[@]}"; do
  d=$root/gpurun_out/$tag/pass$k
  mkdir -p "$d"
  rocprofv3 --pmc $g --output-format csv -d "$d" -o p -- "${args[@]}" > "$d/stdout.log" 2> "$d/stderr.log"
  f=$(find "$d" -name '*counter_collection.csv' | head -n 1)
  [ -n "$f" ] && csvs+=("$f")
  k=$((k+1))
done
srcargs=()
for f in $PMC_SRC; do srcargs+=(--src "$f"); done
[ -n "$PMC_GRID_MAX" ] && srcargs+=(--grid-max "$PMC_GRID_MAX")
python3 "$root/tools/pmc_summary.py" "${srcargs[@]}" "$kernel" "$out" "${csvs[@]}"
