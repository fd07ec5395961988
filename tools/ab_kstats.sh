#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the short bench loop for the tree's library and for variant libraries:
# tools/ab_kstats.sh "tree prebypass" [kernel substring ...]
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for v in $1; do
  if [ $v = tree ]; then unset S3G_LIB_PATH; else export S3G_LIB_PATH=$PWD/s3gaussian_amd/lib/variants/libs3g_$v.so; fi
  d=gpurun_out/kstats_$v
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 0 > $d.log 2>&1
  echo "== $v: $(grep '^{\"metric' $d.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'it/s', d['ms_per_step'], 'ms (under the tracer)')")"
  python tools/kstats.py $d 26
  rm -rf $d
done
