#!/bin/bash
# the two SQ counter passes of collect_profiles.sh on the rasterizer-heavy workload (bench.py --scale-mult 3.5): tools/sq_table.py
cd /tmp && export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
B="python $ROOT/bench.py --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 0 --scale-mult 3.5 --steps 4 --warmup 2"
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
rm -rf /tmp/hv_s1 /tmp/hv_s2
timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d /tmp/hv_s1 -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d /tmp/hv_s2 -- $B > /dev/null 2>&1
python $ROOT/tools/sq_table.py /tmp/hv_s1 /tmp/hv_s2
