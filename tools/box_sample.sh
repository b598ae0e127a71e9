#!/bin/bash
# Runs ON THE GPU BOX: the headline bench lines on one more box of the pool -> gpurun_out/<tag>_bench_{default,fixedbase,msm20,msm17,msm17_async4}_box<k>.json
#   gpurun -- 'bash tools/box_sample.sh r6 2'     then locally: cp gpurun_out/r6_bench_*_box2.json profiles/ ; python tools/box_samples.py r6
TAG=${1:-r6}; K=${2:-2}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_default_box$K.json 2>/dev/null
python bench.py --workload fixedbase --no-cpu-baseline > gpurun_out/${TAG}_bench_fixedbase_box$K.json 2>/dev/null
python bench.py --workload msm --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20_box$K.json 2>/dev/null
python bench.py --workload msm --log2n 17 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17_box$K.json 2>/dev/null
python bench.py --workload msm --log2n 17 --msm-async 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17_async4_box$K.json 2>/dev/null
python tools/box_samples.py $TAG 2>/dev/null | tail -3
