cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
d=gpurun_out/trace_heavy; rm -rf $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python bench.py --steps 12 --warmup 4 --scale-mult 3.5 --no-cpu-baseline --no-alt-paths --no-pmc --no-heavy-raster --sustain-steps 0 > $d.log 2>&1
python tools/step_trace.py $d | grep -E "sort_tiles|bin_kernel|blend|geometry|preprocess|span"
rm -rf $d
