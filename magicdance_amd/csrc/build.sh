#!/bin/bash
# Builds libmagicdance_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libmagicdance_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p "$HERE/build"
pids=()
for f in igemm attention norm elementwise runtime; do
  EXTRA=""
  # attention: keep the MFMA accumulators in VGPRs (gfx950 has one unified file); the softmax touches every S^T / O
  # element each tile, and the AGPR form cost ~5 v_accvgpr moves per MFMA
  if [ "$f" = "attention" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
  ( "$HIPCC" $FLAGS $EXTRA ${MD_EXTRA_FLAGS:-} -c "$HERE/$f.hip" -o "$HERE/build/$f.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$HERE"/build/*.o -o "$OUT"
echo "built $OUT"
