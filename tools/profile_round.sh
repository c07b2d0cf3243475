#!/bin/bash
# Round profile: the rocprofv3 summaries committed under profiles/ (run on the GPU box from the repo root).
#   tools/profile_round.sh r02
cd "$(dirname "$0")/.."
R=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$R
# 1. kernel durations of the default bench run (what --stats prints), as csv
timeout 1200 rocprofv3 --kernel-trace -d gpurun_out/prof_$R/bench -o p -- python3 bench.py --steps 200 --warmup 10 --no-live-traffic > gpurun_out/prof_$R/bench_under_rocprof.json 2> gpurun_out/prof_$R/bench.err </dev/null
python3 tools/rocpd_summary.py gpurun_out/prof_$R/bench/p_results.db "" --csv gpurun_out/prof_$R/${R}_bench_kernel_stats.csv > /dev/null
# 2. the render frame: per-kernel durations + instruction / atomic counters (separate passes)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS -d gpurun_out/prof_$R/raster1 -o p -- python3 tools/probe_raster.py 100000000 5 > gpurun_out/prof_$R/raster1.log 2>&1 </dev/null
timeout 300 rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d gpurun_out/prof_$R/raster2 -o p -- python3 tools/probe_raster.py 100000000 5 > gpurun_out/prof_$R/raster2.log 2>&1 </dev/null
python3 tools/rocpd_summary.py gpurun_out/prof_$R/raster1/p_results.db "" --csv gpurun_out/prof_$R/${R}_raster_frame_kernels_sq.csv > /dev/null
python3 tools/rocpd_summary.py gpurun_out/prof_$R/raster2/p_results.db "" --csv gpurun_out/prof_$R/${R}_raster_frame_kernels_atomics.csv > /dev/null
python3 tools/raster_binding_roof.py gpurun_out/prof_$R/${R}_raster_frame_kernels_sq.csv gpurun_out/prof_$R/${R}_raster_frame_kernels_atomics.csv > gpurun_out/prof_$R/${R}_raster_binding_roof.json
# 3. HBM traffic (FETCH_SIZE / WRITE_SIZE passes)
python3 tools/pmc_traffic.py gpurun_out/prof_$R/${R}_raster_implicit_grid_pmc_traffic.json 3 "raster_,resolve_,hiz_,tile_plan,tile_occ" -- python3 tools/probe_raster.py 100000000 3 > /dev/null 2>&1 </dev/null
python3 tools/pmc_traffic.py gpurun_out/prof_$R/${R}_raster_int32_indices_pmc_traffic.json 3 "raster_,resolve_" -- python3 tools/probe_raster.py 100000000 3 explicit > /dev/null 2>&1 </dev/null
python3 tools/pmc_traffic.py gpurun_out/prof_$R/${R}_project_pmc_kernels.json 3 "project_kernel" -- python3 tools/probe_project.py 100000000 3 f32 > /dev/null 2>&1 </dev/null
python3 - <<PY
import json
d = json.load(open("gpurun_out/prof_$R/${R}_project_pmc_kernels.json"))
k = d["kernels"]["alp::project_kernel<float>"]
json.dump({"kernel": "project_kernel<float>", "vertices_per_launch": 100000000,
           "fetch_bytes_per_launch_x2_corrected": k["fetch_bytes_per_frame_x2_corrected"],
           "write_bytes_per_launch": k["write_bytes_per_frame"], "correction": d["correction"],
           "hbm_bytes_per_launch": k["hbm_bytes_per_frame"], "hbm_bytes_per_vertex": k["hbm_bytes_per_frame"] / 1e8,
           "algorithmic_bytes_per_vertex": 20, "command": d["command"]},
          open("gpurun_out/prof_$R/${R}_project_pmc_traffic.json", "w"), indent=1)
PY
# 4. the frame timeline
tools/probe_frame.sh ${R}_final 100000000 > gpurun_out/prof_$R/${R}_raster_frame_timeline.txt 2>&1
ls gpurun_out/prof_$R
