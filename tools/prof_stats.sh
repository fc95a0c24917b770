#!/bin/bash
# rocprofv3 per-kernel summary of one command (kernel trace + stats only: no counters in the same run).
#   tools/prof_stats.sh <tag> <out.csv> -- python3 <script> [args...]
set -eo pipefail
tag=$1; out=$2; shift 2
[ "$1" == "--" ] && shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
args=()
for a in "$@"; do
  if [ -e "$root/$a" ] && [[ "$a" != /* ]]; then args+=("$root/$a"); else args+=("$a"); fi
done
[[ "$out" != /* ]] && out="$root/$out"
d=$root/gpurun_out/$tag
mkdir -p "$d"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- "${args[@]}" > "$d/stdout.log" 2> "$d/stderr.log"
f=$(find "$d" -name '*kernel_stats.csv' | head -n 1)
cp "$f" "$out"
tail -n 2 "$d/stdout.log" | cut -c1-400
