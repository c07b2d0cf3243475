export TMPDIR=/tmp
mkdir -p gpurun_out/r04e
(time python -m pytest tests/test_gpu_rasterize.py tests/test_gpu_comm.py tests/test_gpu_points.py -x -q) > gpurun_out/r04e/rz_tests.log 2>&1; tail -6 gpurun_out/r04e/rz_tests.log
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r04e/rz -o p -- python3 tools/probe_rasterize.py 100000000 mean median > gpurun_out/r04e/rz.log 2>&1
tail -8 gpurun_out/r04e/rz.log
python3 tools/rocpd_summary.py gpurun_out/r04e/rz/p_results.db "" --csv gpurun_out/r04e/rz_kernels.csv | grep -v "raster_\|resolve\|hiz\|tile_\|surface" | head -40
rm -rf gpurun_out/r04e/rz
python bench.py --steps 20 --no-cpu-baseline --no-raster --no-f64 --no-next-rows --no-dropin > gpurun_out/r04e/bench.json 2> gpurun_out/r04e/bench.err
python -c "
import json;d=json.load(open('gpurun_out/r04e/bench.json'))
print(json.dumps(d['c2_c3_10m']['c3_cma'],indent=1)); print(d['cma']['iters_per_s'], d['cma']['host_ms_per_generation'])"
