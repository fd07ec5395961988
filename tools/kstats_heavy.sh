cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
d=gpurun_out/kstats_heavy
rm -rf $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python bench.py --steps 24 --warmup 4 --scale-mult 3.5 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 0 > $d.log 2>&1
grep '^{"metric' $d.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'it/s', d['ms_per_step'], 'ms'); print(d['config'].get('workload_stats'))"
python tools/kstats.py $d 24
rm -rf $d
