export TMPDIR=/tmp
mkdir -p gpurun_out/r04h
(time python -m pytest tests/test_gpu_rasterize.py tests/test_gpu_surface.py tests/test_gpu_pipeline.py -x -q) > gpurun_out/r04h/tests.log 2>&1; tail -4 gpurun_out/r04h/tests.log
python3 tools/probe_rasterize.py 100000000 mean median max 2>&1 | grep "^mean\|^median\|^max"
python bench.py --steps 50 --no-raster --no-f64 --no-10m --no-cpu-baseline --no-dropin --no-cma > gpurun_out/r04h/next_rows.json 2> gpurun_out/r04h/next_rows.err
python -c "
import json;d=json.load(open('gpurun_out/r04h/next_rows.json'));n=d['next_rows'];print({k:(n[k].get('kernel_ms'),n[k].get('roofline',{}).get('frac')) for k in n})"
