#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
#
# Builds the REFERENCE rasterizer itself (not a restatement) for gfx950 so the parity tests can pin libs3g.so and
# oracle/raster_oracle.c against it on the GPU box:
#
#   /root/reference/submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu
#   + its headers and the vendored glm  --hipify-perl-->  $TMP (never the repo)  --hipcc-->  oracle/_ref/*.so
#
# Outputs (git-ignored, shipped by gpurun like any built .so):
#   oracle/_ref/libref_raster.so       -ffp-contract=off  : one rounding per fp32 op, comparable bit-for-bit with the
#                                                           fp32 oracle on the per-Gaussian geometry and integer state
#   oracle/_ref/libref_raster_fma.so   hipcc default (fast contraction), the analogue of nvcc's default -fmad=true,
#                                      i.e. the arithmetic a user of the reference actually runs
#
# The reference's own build system (setup.py / CMake + torch extension) is not used; oracle/ref_shim.cpp replaces its
# torch wrapper (rasterize_points.cu) with host-pointer C entry points.  Textual fixes applied to the hipified copy only:
#   * `<< <` / `>> >` launch chevrons written with spaces (accepted by nvcc, not by clang)
#   * includes hipify cannot map (`device_launch_parameters.h` -> "", `cooperative_groups/reduce.h`, a cub sub-header)
#   * `__trap()` -> `__builtin_trap()`
# No-op (exit 0) when /root/reference is absent (the GPU box): the prebuilt files are used.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
RAST="${S3G_REFERENCE:-/root/reference}/submodules/depth-diff-gaussian-rasterization"
OUT="$HERE/_ref"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
HIPIFY="${HIPIFY:-/opt/rocm/bin/hipify-perl}"
ARCH="${ARCH:-gfx950}"

if [ ! -d "$RAST/cuda_rasterizer" ]; then
    echo "build_ref: $RAST not present, keeping prebuilt oracle/_ref" >&2
    exit 0
fi
if [ -f "$OUT/libref_raster.so" ] && [ -f "$OUT/libref_raster_fma.so" ] && [ -f "$OUT/libref_knn.so" ] && [ "${1:-}" != "-f" ] \
   && [ "$OUT/libref_raster.so" -nt "$HERE/ref_shim.cpp" ] && [ "$OUT/libref_raster.so" -nt "$HERE/build_ref.sh" ] \
   && [ "$OUT/libref_knn.so" -nt "$HERE/ref_knn_shim.cpp" ] && [ "$OUT/libref_knn.so" -nt "$HERE/build_ref.sh" ]; then
    exit 0
fi

TMP="$(mktemp -d /tmp/s3g_ref_build.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT" "$TMP/cuda_rasterizer"
for f in "$RAST"/cuda_rasterizer/*; do
    b="$(basename "$f")"
    "$HIPIFY" "$f" 2>/dev/null \
      | sed -e '/#include ""/d' -e '/cooperative_groups\/reduce.h/d' -e '/cub\/device\/device_radix_sort.cuh/d' \
            -e 's/__trap()/__builtin_trap()/' -e 's/<< </<<</g' -e 's/>> >/>>>/g' \
      > "$TMP/cuda_rasterizer/${b%.cu}$( [[ "$b" == *.cu ]] && echo .hip || true )"
done
# headers keep their names (the loop above strips only a trailing .cu)
for h in "$RAST"/cuda_rasterizer/*.h; do
    b="$(basename "$h")"; [ -f "$TMP/cuda_rasterizer/$b" ] || { echo "missing $b" >&2; exit 1; }
done

build_one() {   # $1 = output name, rest = extra flags
    local out="$1"; shift
    local objs=()
    for s in forward backward rasterizer_impl; do
        "$HIPCC" --offload-arch="$ARCH" -O3 -std=c++17 -fPIC -w "$@" -I"$RAST/third_party/glm" \
            -c "$TMP/cuda_rasterizer/$s.hip" -o "$TMP/$s.$out.o"
        objs+=("$TMP/$s.$out.o")
    done
    "$HIPCC" --offload-arch="$ARCH" -O3 -std=c++17 -fPIC -w "$@" -I"$TMP/cuda_rasterizer" -I"$RAST/third_party/glm" \
        -x hip -c "$HERE/ref_shim.cpp" -o "$TMP/shim.$out.o"
    "$HIPCC" --offload-arch="$ARCH" -shared -fPIC "${objs[@]}" "$TMP/shim.$out.o" -o "$OUT/$out"
}
build_one libref_raster.so -ffp-contract=off
build_one libref_raster_fma.so

# ---- simple-knn (distCUDA2): submodules/simple-knn/simple_knn.{cu,h} -> rocthrust / hipcub via hipify-perl, same recipe ----------
#   oracle/_ref/libref_knn.so   -ffp-contract=off (squared distances are three products + two adds: comparable bit for bit)
KNN="${S3G_REFERENCE:-/root/reference}/submodules/simple-knn"
mkdir -p "$TMP/knn"
"$HIPIFY" "$KNN/simple_knn.cu" 2>/dev/null \
  | sed -e '/#include ""/d' -e '/cooperative_groups\/reduce.h/d' -e '/device_radix_sort/d' -e '/#define __CUDACC__/d' \
        -e 's/<< </<<</g' -e 's/>> >/>>>/g' > "$TMP/knn/simple_knn.hip"
"$HIPIFY" "$KNN/simple_knn.h" 2>/dev/null > "$TMP/knn/simple_knn.h"
"$HIPCC" --offload-arch="$ARCH" -O3 -std=c++17 -fPIC -w -ffp-contract=off -include cfloat -I"$TMP/knn" -c "$TMP/knn/simple_knn.hip" -o "$TMP/knn/simple_knn.o"
"$HIPCC" --offload-arch="$ARCH" -O3 -std=c++17 -fPIC -w -I"$TMP/knn" -x hip -c "$HERE/ref_knn_shim.cpp" -o "$TMP/knn/shim.o"
"$HIPCC" --offload-arch="$ARCH" -shared -fPIC "$TMP/knn/simple_knn.o" "$TMP/knn/shim.o" -o "$OUT/libref_knn.so"
echo "build_ref: wrote $OUT/libref_raster.so $OUT/libref_raster_fma.so $OUT/libref_knn.so"
