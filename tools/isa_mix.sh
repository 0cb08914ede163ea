#!/bin/bash
# usage: tools/isa_mix.sh <hip source> <mangled-kernel-substring>   (device ISA instruction histogram)
set -e
SRC=$1; PAT=$2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -o /tmp/isa_mix.s "$SRC" --cuda-device-only 2>/dev/null
awk -v pat="$PAT" 'index($0, pat) && /^_Z[^ ]*:/ {f=1} f{print} f&&/^\.Lfunc_end/{exit}' /tmp/isa_mix.s > /tmp/isa_mix_k.s
wc -l /tmp/isa_mix_k.s
grep -E "^\s+[a-z_0-9]+ " /tmp/isa_mix_k.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${3:-30}
