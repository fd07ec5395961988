#!/bin/bash
# Memory-system counters of the HexPlane kernels (address translation, L1 / L2 stalls), to be run ON THE GPU BOX:
#   gpurun --timeout 600 -- 'bash tools/pmc_memsys.sh'
# Two counters per pass and every pass under its own `timeout`: a pass with eight TCP_* counters on tools/hex_probe.py did not
# finish within 13 minutes (round 2) and ate the rest of the GPU budget.  Results: gpurun_out/pmc_memsys.txt.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== pmc_memsys PROBE_ARGS='${PROBE_ARGS:-}' PASSES='${PASSES:-all}'" >> "$OUT/pmc_memsys.txt"
_pass() {   # _pass <tag> <counter> [counter]
  local tag="$1"; shift
  rm -rf "/tmp/pm_$tag"
  if timeout "${PASS_TIMEOUT:-90}" rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "/tmp/pm_$tag" -- \
       python "$ROOT/tools/hex_probe.py" ${PROBE_ARGS:-} > "/tmp/pm_$tag.log" 2>&1; then
    python "$ROOT/tools/pmc_summary.py" "/tmp/pm_$tag" hexplane_ | grep -v -A2 time_rows >> "$OUT/pmc_memsys.txt"
  else
    echo "pass $tag ($*): timed out or failed" >> "$OUT/pmc_memsys.txt"
  fi
}
# PASSES="tlb1 tlb2" selects passes (default: all); PROBE_ARGS="1200000 morton" runs the probe on Morton-ordered points
want() { [ -z "${PASSES:-}" ] || [[ " $PASSES " == *" $1 "* ]]; }
pass() { want "$1" && _pass "$@"; }
pass tlb1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass tlb2 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum
pass l1a  TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
pass l1b  TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pass l1c  TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
pass l2a  TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass l2b  TCC_HIT_sum TCC_MISS_sum
cat "$OUT/pmc_memsys.txt"
