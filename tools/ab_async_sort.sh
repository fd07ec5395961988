#!/bin/bash
# the short bench line with the walk orders refreshed beside the forward pass (default, round 6) against the in-line re-sort every 16th backward
for v in ${1:-1 0 1 0}; do
  echo "== bench, S3G_HEX_ASYNC_SORT=$v"
  S3G_HEX_ASYNC_SORT=$v python bench.py --steps 32 --warmup 6 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps ${SUSTAIN:-160} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'sustained', d.get('sustained_iters_per_s'), 'p99', d.get('step_ms_p99'), 'instrumented', d.get('instrumented_loop_ms_per_step'), 'render', d.get('render_ms_per_frame'), ' '.join(k['kernel'].split('::')[1].replace('_kernel','')+'='+str(k['avg_launch_ms']) for k in d['roofline']['kernels']))"
done
