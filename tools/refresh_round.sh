#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole evidence set of a round at one build.
#   gpurun -- 'bash tools/refresh_round.sh r3'     then locally: bash tools/collect_round.sh r3
set -e
TAG=${1:-r6}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8        # the host-buffer pipeline beside torch's streams (include/jubjub_hip.h: the application sets it, not the library)
mkdir -p gpurun_out
python tools/profile_round.py --tag $TAG > gpurun_out/${TAG}_profile.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python bench.py --workload decompress --decompress-flags 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_dec1.json 2>/dev/null
python bench.py --workload decompress --decompress-flags 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_dec3.json 2>/dev/null
python bench.py --workload msm --log2n 22 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm22.json 2>/dev/null
python bench.py --workload msm --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20.json 2>/dev/null      # without the profiler's per-dispatch overhead
python bench.py --workload msm --log2n 17 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17.json 2>/dev/null
python bench.py --workload msm --log2n 18 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm18.json 2>/dev/null
python bench.py --workload msm --log2n 19 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm19.json 2>/dev/null
python bench.py --workload msm --msm-async 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20_async2.json 2>/dev/null   # two jobs in flight over the context's two lanes (jj_msm_begin / _finish)
python bench.py --workload msm --log2n 17 --msm-async 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17_async2.json 2>/dev/null
python bench.py --workload msm --log2n 17 --msm-async 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17_async4.json 2>/dev/null
python bench.py --workload msm --msm-async 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20_async4.json 2>/dev/null
python bench.py --workload msm --log2n 17 --msm-contexts 2 --msm-async 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17_ctx2.json 2>/dev/null   # two contexts x two jobs in flight: sustained throughput
python bench.py --workload msm --log2n 17 --msm-contexts 4 --msm-async 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm17_ctx4.json 2>/dev/null
python bench.py --workload msm --msm-contexts 2 --msm-async 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20_ctx2.json 2>/dev/null
python bench.py --workload msm --log2n 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm10.json 2>/dev/null           # small-batch path
python bench.py --workload fixedbase --fb-window 16 --no-cpu-baseline > gpurun_out/${TAG}_bench_fb16.json 2>/dev/null
python bench.py --workload fixedbase --fb-window 6 --no-cpu-baseline > gpurun_out/${TAG}_bench_fb6.json 2>/dev/null      # round 2's kernel: signed 6-bit windows
bash tools/msm_profile.sh $TAG 17 > gpurun_out/${TAG}_msm17_profile.log 2>&1                                             # gpurun_out/<tag>_msm17_kernel_stats.txt
python tools/latency.py > gpurun_out/${TAG}_latency.txt 2>&1
(bash tools/msm_timeline.sh 20; bash tools/msm_timeline.sh 17; bash tools/msm_timeline.sh 10) > gpurun_out/${TAG}_msm_timeline.txt 2>&1            # start / end of every kernel of one call
python tools/composite_bench.py 22 > gpurun_out/${TAG}_fixedbase_composite.txt 2>&1
(python experiments/misc/msm_concurrency.py 17 60; python experiments/misc/msm_concurrency.py 20 30; python experiments/misc/msm_concurrency.py 10 200) > gpurun_out/${TAG}_msm_concurrency.txt 2>&1
python experiments/misc/msm_partition_cost.py 20 8 > gpurun_out/${TAG}_msm_partition_cost.txt 2>&1
(python tests/config1_cpu.py; lscpu | grep -E "^CPU\(s\)|Model name") > gpurun_out/${TAG}_config1_cpu.txt 2>&1
# round 4: the host-pointer (drop-in) path priced against the host link, jj_multi_* on one GPU listed once / twice, the device-side MSM finish,
# the LDS counters of the fixed-base select, the LDS / energy probes
bash tools/pcie_inclusive.sh $TAG > /dev/null 2>&1
python tools/multi_bench.py > gpurun_out/${TAG}_multi_bench.txt 2>&1
python tools/msm_dev_finish.py > gpurun_out/${TAG}_msm_dev_finish.txt 2>&1
bash tools/fixedbase_floor.sh > gpurun_out/${TAG}_fixedbase_select_pmc.txt 2>&1          # round 5: the comb with two / one / no shuffle round (probe libraries built on the CPU: python tools/fixedbase_floor.py build)
bash tools/stall_pmc.sh > gpurun_out/${TAG}_stall_pmc.txt 2>&1
[ -x experiments/lds_probe/probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o experiments/lds_probe/probe experiments/lds_probe/probe.hip
[ -x experiments/lds_probe/energy_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o experiments/lds_probe/energy_probe experiments/lds_probe/energy_probe.hip
(./experiments/lds_probe/energy_probe; ./experiments/lds_probe/probe) > gpurun_out/${TAG}_issue_energy_probe.txt 2>&1
JJ_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --workload msm --msm-exchange c --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20_rccl1.json 2>/dev/null </dev/null   # one rank over RCCL, the exchange behind the C ABI
python bench.py --workload msm > gpurun_out/${TAG}_bench_msm20_cpu.json 2>/dev/null                                                   # with the CPU baseline: naive fold + bucket method
(python tests/host_tail_time.py 1 8; python tests/host_tail_time.py scalar 1 8) > gpurun_out/${TAG}_host_tail.txt 2>&1     # MSM host tail on the box's CPU: AVX-512 IFMA chain | scalar chain
for m in 0 1 0 1; do python bench.py --opt host_tail_scalar=$m --workload msm --log2n 17 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host_tail_scalar=$m  msm 2^17: %.4f ms per call (kernels %.4f ms), verified %s' % (d['config']['ms_per_pass'], d['roofline']['kernel_ms'], d['verified']))"; done >> gpurun_out/${TAG}_host_tail.txt 2>&1
timeout 300 python tests/soak_host.py 120 > gpurun_out/${TAG}_soak_host.txt 2>&1 || echo "HOST SOAK FAILED" >> gpurun_out/${TAG}_soak_host.txt
timeout 600 python tests/soak.py 240 3000 > gpurun_out/${TAG}_soak.txt 2>&1 || echo "SOAK FAILED" >> gpurun_out/${TAG}_soak.txt
# round 5: constant-time ladder window widths against the table ladder; the two-level bucket reduce's sweep; the stand-alone fault reproducer
python experiments/misc/vb_ct_window.py > gpurun_out/${TAG}_vb_ct_window.txt 2>&1
python experiments/misc/msm_reduce_l1_sweep.py 18 19 20 21 22 > gpurun_out/${TAG}_msm_reduce_l1_sweep.txt 2>&1
python experiments/misc/msm_sort_hist_ab.py > gpurun_out/${TAG}_msm_sort_hist_ab.txt 2>&1
python experiments/misc/msm_allgather_pipeline.py 20 8 > gpurun_out/${TAG}_msm_allgather_pipeline.txt 2>&1     # one rank of eight with the other ranks played by tools/loopback_comm.cpp: synchronous jj_msm_allgather against jj_msm_allgather_begin jobs in flight
JJ_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --workload msm --msm-exchange c --msm-async 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_msm20_rccl1_async4.json 2>/dev/null </dev/null   # the real ncclAllGather (one rank) on the jobs' lanes
# round 6: clock and issue rate of the roofline denominator beside the ladder; the 2^17-term MSM's two-launch front end and the
# LDS-staged bucket offsets against the round-5 forms (alternating bench runs); window count at 2^18 / 2^19 terms; the jobs-in-flight soak
bash tools/peak_clock.sh > gpurun_out/${TAG}_peak_clock.log 2>&1; cp gpurun_out/peak_clock.txt gpurun_out/${TAG}_peak_clock.txt
for o in "msm_front1=1" "msm_front1=0" "msm_front1=1" "msm_front1=0" "msm_acc_lds=1" "msm_acc_lds=0" "msm_acc_lds=1" "msm_acc_lds=0"; do python bench.py --opt $o --workload msm --log2n 17 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o  msm 2^17: %.4f ms per call, verified %s' % (d['config']['ms_per_pass'], d['verified']))"; done > gpurun_out/${TAG}_msm17_ab.txt 2>&1
for lg in 18 19; do for w in 16 17 16 17; do python bench.py --opt msm_windows=$w --workload msm --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('msm_windows=$w  msm 2^$lg: %.4f ms per call, verified %s' % (d['config']['ms_per_pass'], d['verified']))"; done; done > gpurun_out/${TAG}_msm_windows_ab.txt 2>&1
timeout 400 python tests/soak_jobs.py 180 9000 > gpurun_out/${TAG}_soak_jobs.txt 2>&1 || echo "JOBS SOAK FAILED" >> gpurun_out/${TAG}_soak_jobs.txt
[ -x experiments/mad_banks/probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o experiments/mad_banks/probe experiments/mad_banks/probe.hip 2>/dev/null
[ -x experiments/sync_latency/probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o experiments/sync_latency/probe experiments/sync_latency/probe.hip 2>/dev/null
./experiments/mad_banks/probe > gpurun_out/${TAG}_mad_banks.txt 2>&1          # multiply-add issue rate against VGPR banks, operand kinds and waves per SIMD
./experiments/sync_latency/probe > gpurun_out/${TAG}_sync_latency.txt 2>&1    # what the host's wait for a kernel costs: event, stream, flag in host memory
for N in 17 18 20; do for L in 2 3 4; do for A in 2 4 6; do python bench.py --workload msm --log2n $N --msm-async $A --opt msm_lanes=$L --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$N lanes $L async $A: %.4f ms per MSM, frac %.3f, verified %s' % (d['config']['ms_per_pass'], d['roofline']['frac'], d['verified']))"; done; done; done > gpurun_out/${TAG}_msm_lanes.txt 2>&1
tail -1 gpurun_out/${TAG}_profile.log
