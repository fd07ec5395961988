#!/bin/bash
# Final evidence of a round, run ON THE GPU BOX (gpurun): bench line, kernel stats, PMC traffic, sync gap, 2-rank functional run.
#   gpurun -- 'bash tools/collect_profiles.sh r02'
# Everything lands under gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -uo pipefail
TAG="${1:-rNN}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-alt-paths > /dev/null 2>&1
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_bench_kernel_stats.csv"
python "$ROOT/tools/sync_gap.py" "$OUT/prof_stats" > "$OUT/${TAG}_sync_gap.txt" 2>&1
python "$ROOT/tools/step_trace.py" "$OUT/prof_stats" > "$OUT/${TAG}_step_trace.txt" 2>&1
rm -rf "$OUT/prof_stats"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-alt-paths > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-alt-paths > /dev/null 2>&1
python "$ROOT/tools/pmc_traffic.py" /tmp/pmc_f /tmp/pmc_w "$OUT/${TAG}_pmc" > "$OUT/${TAG}_pmc_traffic.txt" 2>&1
for k in f w; do c=$(find /tmp/pmc_$k -name "*counter_collection.csv" | head -1); [ -n "$c" ] && cp "$c" "$OUT/${TAG}_pmc/raw_$k.csv"; done
# the bench line last, so that its `traffic` fields are the PMC bytes collected a minute earlier on this very box
cp "$OUT/${TAG}_pmc/kernel_traffic.json" "$ROOT/profiles/kernel_traffic.json" 2>/dev/null
cd /tmp
python "$ROOT/bench.py" > "$OUT/${TAG}_bench_line.json" 2> "$OUT/${TAG}_bench_line.err"
# R sweep (same Gaussians, every scale multiplied: more (tile, Gaussian) instances per view) and P sweep, final tree
: > "$OUT/${TAG}_sweep.jsonl"
for extra in "--scale-mult 2.0" "--scale-mult 3.5" "--P 600000" "--P 2500000"; do
  python "$ROOT/bench.py" --no-cpu-baseline --no-alt-paths $extra >> "$OUT/${TAG}_sweep.jsonl" 2>/dev/null
done
S3G_HEX_BACKWARD=walk python "$ROOT/bench.py" --no-cpu-baseline --no-alt-paths --P 2500000 >> "$OUT/${TAG}_sweep.jsonl" 2>/dev/null
# two ranks on the one GPU of the lease: functional check of the view-parallel path (gloo; RCCL refuses two ranks on one device)
cd "$ROOT"
S3G_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 3 --warmup 1 --P 300000 > "$OUT/${TAG}_two_ranks_gloo.json" 2> "$OUT/${TAG}_two_ranks_gloo.err"
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
    bench.py --gpus 2 --steps 2 --warmup 1 --P 100000 > "$OUT/${TAG}_two_ranks_rccl.json" 2> "$OUT/${TAG}_two_ranks_rccl.err"
tail -2 "$OUT/${TAG}_two_ranks_rccl.err" > "$OUT/${TAG}_two_ranks_rccl_tail.txt"
echo done
