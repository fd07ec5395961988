python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warn | tail -4 > gpurun_out/r06_final_gpu_suite.txt; cat gpurun_out/r06_final_gpu_suite.txt
python bench.py > gpurun_out/r06_bench_line_final.json 2> gpurun_out/r06_bench_line_final.err
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_line_final.json')); p=d['config']['paths']; print(d['value'], d['ms_per_step'], d['sustained_iters_per_s'], p['heavy_raster']['ms_per_step'], p['patched']['ms_per_step'], {k:v for k,v in d.items() if 'render' in k})"
