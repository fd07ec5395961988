for v in new prebypass new prebypass; do
  if [ $v = new ]; then unset S3G_LIB_PATH; else export S3G_LIB_PATH=$PWD/s3gaussian_amd/lib/variants/libs3g_$v.so; fi
  echo "== $v"
  python tools/scatter_context_probe.py 1200000 32 6 2>&1 | grep "^[0AG]\."
  python bench.py --steps 32 --warmup 6 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 160 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], 'sustained', d.get('sustained_iters_per_s'), 'instrumented', d.get('instrumented_loop_ms_per_step'), ' '.join(k['kernel'].split('::')[1].replace('_kernel','')+'='+str(k['avg_launch_ms']) for k in d['roofline']['kernels'] if 'hexplane' in k['kernel']))"
done
