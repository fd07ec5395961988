#!/bin/bash
# Final evidence of a round, run ON THE GPU BOX (gpurun): bench line, kernel stats, PMC traffic + SQ counter passes, sync gap,
# 2-rank functional run.      gpurun -- 'bash tools/collect_profiles.sh r04'
# Everything lands under gpurun_out/<tag>_*; copy what should be judged into profiles/.  Counters are collected in their own
# passes with --kernel-trace only (never together with --sys-trace / hip / hsa domains).
set -uo pipefail
TAG="${1:-rNN}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-alt-paths --sustain-steps 0"
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
TCC="TCC_ATOMIC TCC_REQ TCC_HIT TCC_MISS"
# 1. kernel trace of the bench: per-kernel totals, one step launch by launch, idle time around the (former) host sync
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- $B --steps 20 --warmup 3 > /dev/null 2>&1
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_bench_kernel_stats.csv"
python "$ROOT/tools/sync_gap.py" "$OUT/prof_stats" > "$OUT/${TAG}_sync_gap.txt" 2>&1
python "$ROOT/tools/step_trace.py" "$OUT/prof_stats" > "$OUT/${TAG}_step_trace.txt" 2>&1
rm -rf "$OUT/prof_stats"
# 2. HBM traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass) + the two SQ passes of the same command
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d /tmp/pmc_s1 -- $B --steps 4 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d /tmp/pmc_s2 -- $B --steps 4 --warmup 2 > /dev/null 2>&1
python "$ROOT/tools/pmc_traffic.py" /tmp/pmc_f /tmp/pmc_w "$OUT/${TAG}_pmc" /tmp/pmc_s1 > "$OUT/${TAG}_pmc_traffic.txt" 2>&1
( echo "# rocprofv3 --pmc <SQ counters> --kernel-trace -- python bench.py --no-cpu-baseline --no-alt-paths --steps 4 --warmup 2  (two passes; tools/sq_table.py; tree of round ${TAG})"
  python "$ROOT/tools/sq_table.py" /tmp/pmc_s1 /tmp/pmc_s2 ) > "$OUT/${TAG}_bench_sq_pmc.txt" 2>&1
# 3. the sampler alone (tools/hex_probe.py): SQ passes + L2 request / atomic counters
timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d /tmp/hex_s1 -- python "$ROOT/tools/hex_probe.py" > /dev/null 2>&1
timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d /tmp/hex_s2 -- python "$ROOT/tools/hex_probe.py" > /dev/null 2>&1
timeout 200 rocprofv3 --pmc $TCC --kernel-trace --output-format csv -d /tmp/hex_t -- python "$ROOT/tools/hex_probe.py" > /dev/null 2>&1
( echo "# rocprofv3 --pmc <SQ / TCC counters> --kernel-trace -- python tools/hex_probe.py  (three passes; tools/sq_table.py; 1.2 M points; tree of round ${TAG})"
  python "$ROOT/tools/sq_table.py" /tmp/hex_s1 /tmp/hex_s2 /tmp/hex_t ) > "$OUT/${TAG}_hexplane_sq_pmc.txt" 2>&1
# 3b. the twelve scatter walks one by one (VERDICT r5 next #2 iii)
timeout 300 python "$ROOT/tools/hex_probe.py" 1200000 walks > "$OUT/${TAG}_hex_walks.txt" 2>&1
# 4. the bench line last, so that its `traffic` / VALU fields are the PMC data collected a minute earlier on this very box
cp "$OUT/${TAG}_pmc/kernel_traffic.json" "$ROOT/profiles/kernel_traffic.json" 2>/dev/null
cd /tmp
python "$ROOT/bench.py" > "$OUT/${TAG}_bench_line.json" 2> "$OUT/${TAG}_bench_line.err"
# 5. R sweep (same Gaussians, every scale multiplied) and P sweep
: > "$OUT/${TAG}_sweep.jsonl"
for extra in "--scale-mult 3.5" "--P 600000" "--P 2500000" "--sync-raster"; do
  timeout 200 $B $extra >> "$OUT/${TAG}_sweep.jsonl" 2>/dev/null
done
# 6. two ranks on the one GPU of the lease through the PLAIN command line (bench.py starts its own ranks; gloo, because RCCL
#    refuses two ranks on one device -- functional check of the view-parallel path, not a scaling number)
cd "$ROOT"
S3G_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --P 300000 > "$OUT/${TAG}_two_ranks_gloo.json" 2> "$OUT/${TAG}_two_ranks_gloo.err"
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 --P 100000 > "$OUT/${TAG}_two_ranks_refused.json" 2> "$OUT/${TAG}_two_ranks_refused.err"
echo "exit code of the plain --gpus 2 command on a 1-GPU box without the gloo override: $?" >> "$OUT/${TAG}_two_ranks_refused.err"
# 7. RCCL itself, as a process group of ONE rank (S3G_FORCE_DIST=1, dp.force_dist): the `comm` block of an executed nccl backend
S3G_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --sustain-steps 0 \
  > "$OUT/${TAG}_rccl_one_rank_line.json" 2> "$OUT/${TAG}_rccl_one_rank_line.err"
echo done
