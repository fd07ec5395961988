#!/bin/bash
# like ab_kstats.sh on the heavy-raster workload (--scale-mult 3.5) AND the headline workload: $1 = variants ("tree" = the tree's library)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for v in $1; do
  if [ $v = tree ]; then unset S3G_LIB_PATH; else export S3G_LIB_PATH=$PWD/s3gaussian_amd/lib/variants/libs3g_$v.so; fi
  for mult in 3.5 1.0; do
    d=gpurun_out/kstats_${v}_$mult
    rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python bench.py --steps 24 --warmup 4 --scale-mult $mult --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 0 > $d.log 2>&1
    echo "== $v, scale x $mult: $(grep '^{"metric' $d.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'it/s', d['ms_per_step'], 'ms (under the tracer)')")"
    python tools/kstats.py $d 40 | grep -E "${2:-sort_tiles}"
    rm -rf $d $d.log
  done
done
