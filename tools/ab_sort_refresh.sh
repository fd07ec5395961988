#!/bin/bash
# bench line (short form) under several re-sort periods of the HexPlane walk orders: $1 = list of S3G_HEX_SORT_REFRESH values
for r in $1; do
  echo "== bench, S3G_HEX_SORT_REFRESH=$r"
  S3G_HEX_SORT_REFRESH=$r python bench.py --steps 32 --warmup 6 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps ${SUSTAIN:-160} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'sustained', d.get('sustained_iters_per_s'), 'instrumented', d.get('instrumented_loop_ms_per_step'), ' '.join(k['kernel'].split('::')[1].replace('_kernel','')+'='+str(k['avg_launch_ms']) for k in d['roofline']['kernels'] if 'hexplane' in k['kernel']))"
done
