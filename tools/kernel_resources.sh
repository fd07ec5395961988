#!/bin/bash
# tools/kernel_resources.sh [name.o ...]: VGPR / SGPR / LDS / scratch of every gfx950 kernel in s3gaussian_amd/lib/*.o
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; LL=/opt/rocm/lib/llvm/bin; T="$(mktemp -d)"; trap 'rm -rf "$T"' EXIT
for o in "${@:-$ROOT/s3gaussian_amd/lib/*.o}"; do for f in $o; do
  $LL/llvm-objcopy --dump-section .hip_fatbin=$T/fb "$f" 2>/dev/null || continue
  $LL/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb --output=$T/co --unbundle
  $LL/llvm-readelf --notes $T/co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.group_segment_fixed_size:/{l=$2} /\.private_segment_fixed_size:/{p=$2} /\.agpr_count:/{a=$2} /\.wavefront_size:/{printf "%-90s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %s\n", substr(n,1,90), v, a, s, l, p}'
done; done
