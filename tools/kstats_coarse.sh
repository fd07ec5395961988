cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
python tools/coarse_probe.py 40 2>&1 | grep coarse
d=gpurun_out/kstats_coarse; rm -rf $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python tools/coarse_probe.py 40 > $d.log 2>&1
python tools/kstats.py $d 22
rm -rf $d
