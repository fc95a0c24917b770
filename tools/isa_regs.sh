#!/bin/bash
# registers / scratch / occupancy per kernel of one .hip file:  tools/isa_regs.sh laughter-detection-icsi_amd/csrc/conv_b3.hip [filter]
src=$1; filt=${2:-.}
out=/tmp/$(basename "$src" .hip).s
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -I "$(dirname "$0")/../include" -S --cuda-device-only -o "$out" "$src" 2>/dev/null || exit 1
awk '/^_Z[^ ]*: / {name=$1; sub(/:$/, "", name)} /; NumVgprs:/ {v=$3} /; ScratchSize:/ {s=$3} /; Occupancy:/ {print name, "vgpr", v, "scratch", s, "occ", $3}' "$out" | c++filt | sed 's/(anonymous namespace):://g; s/(unsigned char const\*.*) vgpr/ vgpr/; s/^void //' | grep -E "$filt"
