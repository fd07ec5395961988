python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
run() { python bench.py --steps 32 --warmup 6 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'sustained', d.get('sustained_iters_per_s'), 'instrumented', d.get('instrumented_loop_ms_per_step'), ' '.join(k['kernel'].split('::')[1].replace('_kernel','')+'='+str(k['avg_launch_ms']) for k in d['roofline']['kernels']))"; }
echo "== off"; S3G_HEX_ASYNC_SORT=0 run
echo "== post"; S3G_HEX_ASYNC_WHERE=post run
echo "== post, prio 0"; S3G_HEX_ASYNC_WHERE=post S3G_HEX_ASYNC_PRIO=0 run
echo "== post, prio 1"; S3G_HEX_ASYNC_WHERE=post S3G_HEX_ASYNC_PRIO=1 run
echo "== post, prio -1"; S3G_HEX_ASYNC_WHERE=post S3G_HEX_ASYNC_PRIO=-1 run
echo "== off"; S3G_HEX_ASYNC_SORT=0 run
echo "== post"; S3G_HEX_ASYNC_WHERE=post run
