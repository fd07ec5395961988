#!/bin/bash
# per-kernel averages of one short bench.py run under rocprofv3:  tools/kstats_run.sh "<bench.py arguments>" [rows]
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
d=gpurun_out/kstats_run
rm -rf $d
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 0 $1 > $d.log 2>&1
echo "== bench.py $1: $(grep '^{"metric' $d.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'it/s', d['ms_per_step'], 'ms (under the tracer); render', d['config'].get('render_ms_per_frame'))")"
python tools/kstats.py $d ${2:-28}
rm -rf $d
