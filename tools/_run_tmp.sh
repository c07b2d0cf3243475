mkdir -p gpurun_out/r04f
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r04f/gpu_tests.log 2>&1; tail -3 gpurun_out/r04f/gpu_tests.log
(time python bench.py) > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err; tail -3 gpurun_out/r04f/bench.err
python -c "
import json;d=json.load(open('gpurun_out/r04f/bench.json'))
print(json.dumps(d['roofline'],indent=0)); print(json.dumps(d['next_rows']['f2_rasterize_points'],indent=1)); print(json.dumps(d['f2_device_fed'],indent=1)); print(d['pipeline_full_size']['stage_ms'])"
