#!/bin/bash
# tools/isa_stats.sh name.o [kernel-substring]: static instruction mix (VALU / packed / SALU / memory) per gfx950 kernel
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; LL=/opt/rocm/lib/llvm/bin; T="$(mktemp -d)"; trap 'rm -rf "$T"' EXIT
$LL/llvm-objcopy --dump-section .hip_fatbin=$T/fb "$ROOT/s3gaussian_amd/lib/$1" && \
$LL/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb --output=$T/co --unbundle && \
$LL/llvm-objdump -d $T/co > ${ISA_OUT:-$T/s}
python3 - "${ISA_OUT:-$T/s}" "${2:-}" <<'PY'
import re, sys, collections
cur = None; stats = collections.defaultdict(collections.Counter)
for line in open(sys.argv[1]):
    m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
    if m: cur = m.group(1); continue
    m = re.match(r'^\s+(\S+)\s', line)
    if m and cur: stats[cur][m.group(1)] += 1
for k, c in stats.items():
    if sys.argv[2] and sys.argv[2] not in k: continue
    tot = sum(c.values()); f = lambda pre: sum(v for o, v in c.items() if o.startswith(pre))
    print(f"{k[:70]:70s} total {tot:5d} valu {f('v_'):5d} (pk {f('v_pk'):4d} mov {c['v_mov_b32_e32']:4d}) salu {f('s_'):5d} mem {f(('global','ds_','buffer','flat','scratch')):4d} scratch {f('scratch'):3d}")
    if sys.argv[2]: print('   ', c.most_common(24))
PY
