#!/bin/bash
# tools/isa.sh <file.hip> <mangled-kernel-substring> [out.s]: device ISA of one kernel of a translation unit
# (hipcc -S --cuda-device-only), cut out of the whole-file listing.  Prints register / scratch figures.
set -e
SRC=$1; PAT=$2; OUT=${3:-/tmp/kernel.s}
cd "$(dirname "$0")/../mmhand_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm \
    -S --cuda-device-only -o /tmp/_all.s "$SRC" 2>&1 | grep -v "warning\|^$" || true
NAME=$(grep -o "^_Z[A-Za-z0-9_]*${PAT}[A-Za-z0-9_]*:" /tmp/_all.s | head -1 | tr -d ':')
[ -n "$NAME" ] || { echo "no kernel matching $PAT"; exit 1; }
L=$(grep -n "^${NAME}:" /tmp/_all.s | cut -d: -f1)
E=$(grep -n "\.amdhsa_kernel ${NAME}\$" /tmp/_all.s | cut -d: -f1)
awk -v s=$L -v e=$((E+45)) 'NR>=s && NR<=e' /tmp/_all.s > "$OUT"
echo "$NAME -> $OUT ($(wc -l < "$OUT") lines)"
grep -E "amdhsa_next_free_vgpr|amdhsa_accum_offset|amdhsa_private_segment_fixed_size|amdhsa_next_free_sgpr" "$OUT" || true
echo "scratch ops: $(grep -c 'scratch_' "$OUT")  accvgpr copies: $(grep -c 'v_accvgpr' "$OUT")  mfma: $(grep -c 'v_mfma' "$OUT")"
