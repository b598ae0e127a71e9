#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole evidence set of a round at one build.
#   gpurun -- 'bash tools/refresh_round.sh r2'     then locally: bash tools/collect_round.sh r2
set -e
TAG=${1:-r2}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/profile_round.py --tag $TAG > gpurun_out/${TAG}_profile.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python bench.py --workload decompress --decompress-flags 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_dec1.json 2>/dev/null
python bench.py --workload decompress --decompress-flags 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_dec3.json 2>/dev/null
python bench.py --workload msm --log2n 22 --no-cpu-baseline --no-verify > gpurun_out/${TAG}_bench_msm22.json 2>/dev/null
python bench.py --workload msm --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20.json 2>/dev/null      # without the profiler's per-dispatch overhead
python bench.py --workload msm --log2n 17 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17.json 2>/dev/null
python bench.py --workload fixedbase --fb-window 16 --no-cpu-baseline > gpurun_out/${TAG}_bench_fb16.json 2>/dev/null
python tools/latency.py > gpurun_out/${TAG}_latency.txt 2>&1
timeout 600 python tests/soak.py 240 2000 > gpurun_out/${TAG}_soak.txt 2>&1 || echo "SOAK FAILED" >> gpurun_out/${TAG}_soak.txt
tail -1 gpurun_out/${TAG}_profile.log
