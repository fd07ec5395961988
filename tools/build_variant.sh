#!/bin/bash
# tools/build_variant.sh NAME 'sed-expression' FILE.hip  -- builds s3gaussian_amd/lib/variants/libs3g_NAME.so from the current
# tree with one source file patched by a sed expression (kernel A/B experiments: run with S3G_LIB_PATH=... on the GPU box).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; EXPR="$2"; FILE="$3"
SRC="$ROOT/s3gaussian_amd/csrc"; OUT="$ROOT/s3gaussian_amd/lib/variants"; TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$SRC"
objs=()
for f in "$SRC"/*.hip; do
  b="$(basename "$f" .hip)"
  if [ "$b.hip" = "$FILE" ]; then
    sed -e "$EXPR" "$f" > "$SRC/_variant_$b.hip"
    /opt/rocm/bin/hipcc $FLAGS -c "$SRC/_variant_$b.hip" -o "$TMP/$b.o"; rm -f "$SRC/_variant_$b.hip"
    objs+=("$TMP/$b.o")
  else
    objs+=("$ROOT/s3gaussian_amd/lib/$b.o")
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT/libs3g_$NAME.so"
echo "$OUT/libs3g_$NAME.so"
