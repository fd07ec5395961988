#!/bin/bash
# A/B of the HexPlane backward's experiment switches (S3G_HEX_VARIANT, csrc/hexplane.hip::g_hex_variant) on ONE box:
# the sampler probe, the sampler's GPU tests under the non-default variants, and the short bench line.
# usage: tools/ab_hex_variant.sh "0 1 2 4 6 8 14" "6 14" "0 6 14 0 6"
mkdir -p gpurun_out
out=gpurun_out/${TAG:-r06_hex_variant}.txt
: > $out
for v in $1; do
  echo "== probe, variant $v" >> $out
  S3G_HEX_VARIANT=$v python tools/hex_probe.py 1200000 2>/dev/null | tail -5 >> $out
done
for v in $2; do
  echo "== tests/test_hexplane_gpu.py + test_parity_fullsize_gpu.py, variant $v" >> $out
  S3G_HEX_VARIANT=$v python -m pytest tests/test_hexplane_gpu.py tests/test_parity_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $out
done
for v in $3; do
  echo "== bench, variant $v" >> $out
  S3G_HEX_VARIANT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'sustained', d.get('sustained_iters_per_s'), ' '.join(k['kernel'].split('::')[1].replace('_kernel','')+'='+str(k['avg_launch_ms']) for k in d['roofline']['kernels']))" >> $out
done
cat $out
