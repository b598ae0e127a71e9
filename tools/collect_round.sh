#!/bin/bash
# Runs LOCALLY after tools/refresh_round.sh came back through gpurun_out/: copies the round's evidence into profiles/.
set -e
TAG=${1:-r6}
cd "$(dirname "$0")/.."
for f in gpurun_out/profiles_$TAG/*; do
  b=$(basename $f)
  case $b in ${TAG}_*_kernel_stats.txt|${TAG}_*_pmc.txt|${TAG}_*_bench.json|traffic.json|${TAG}_kernel_resources.txt|${TAG}_instr_mix.txt) cp $f profiles/;; esac
done
for f in default dec1 dec3 msm20 msm22 msm17 msm18 msm19 msm20_async2 msm17_async2 msm17_async4 msm20_async4 msm17_ctx2 msm17_ctx4 msm20_ctx2 msm10 fb16 fb6; do cp gpurun_out/${TAG}_bench_$f.json profiles/${TAG}_bench_$f.json; done
BID=$(python3 -c "import json; print(json.load(open('profiles/traffic.json'))['build_id'])")
HDR="# commit $(cat profiles/BUILD_COMMIT) | build_id $BID | MI355X gfx950"
cp gpurun_out/${TAG}_msm17_kernel_stats.txt profiles/${TAG}_msm17_kernel_stats.txt
(echo "$HDR"; echo "# command: python tools/latency.py   (median wall time per C-ABI call, device-resident inputs)"; grep -v amdgpu gpurun_out/${TAG}_latency.txt) > profiles/${TAG}_latency.txt
(echo "$HDR"; echo "# command: bash tools/msm_timeline.sh 20 | 17 | 10   (rocprofv3 --kernel-trace of a short bench run; the kernels of the last MSM call)"; grep -v amdgpu gpurun_out/${TAG}_msm_timeline.txt) > profiles/${TAG}_msm_timeline.txt
(echo "$HDR"; echo "# command: python tools/composite_bench.py 22"; grep -v amdgpu gpurun_out/${TAG}_fixedbase_composite.txt) > profiles/${TAG}_fixedbase_composite.txt
(echo "$HDR"; echo "# command: python experiments/misc/msm_concurrency.py <log2n> <iters> for 2^17, 2^20, 2^10 terms   (K contexts = K streams + K workspace sets on ONE GPU, one host thread each; sync = jj_msm per call, async2 = two jobs in flight per context; best of 3)"; grep -v amdgpu gpurun_out/${TAG}_msm_concurrency.txt) > profiles/${TAG}_msm_concurrency.txt
(echo "$HDR"; echo "# command: python experiments/misc/msm_partition_cost.py 20 8"; grep -v amdgpu gpurun_out/${TAG}_msm_partition_cost.txt) > profiles/${TAG}_msm_partition_cost.txt
(echo "$HDR"; echo "# command: python tests/config1_cpu.py   (BASELINE.json configs[0] on the CPU port of the reference algorithm, host of the GPU box)"; cat gpurun_out/${TAG}_config1_cpu.txt) > profiles/${TAG}_config1_cpu.txt
(echo "$HDR"; echo "# command: python tests/soak.py 240 3000   (randomised differential soak of every entry point against the C oracle; last rounds and verdict)"; grep -v amdgpu gpurun_out/${TAG}_soak.txt | tail -6) > profiles/${TAG}_soak.txt
# round 4 additions
for f in gpurun_out/${TAG}_bench_host_*.json gpurun_out/${TAG}_bench_msm20_rccl1.json gpurun_out/${TAG}_bench_msm20_cpu.json; do [ -s "$f" ] && cp $f profiles/; done
(echo "$HDR"; echo "# command: ./tools/pcie_probe   (tools/pcie_probe.cpp: what the host link of the box gives; the host-pointer path is priced against it)"; cat gpurun_out/${TAG}_pcie_probe.txt) > profiles/${TAG}_pcie_probe.txt
(echo "$HDR"; echo "# command: bash tools/pcie_inclusive.sh $TAG   (bench.py --host-buffers pinned|pageable at config sizes: the C-ABI call on HOST arrays is the timed region; full lines in ${TAG}_bench_host_*.json)"; cat gpurun_out/${TAG}_pcie_inclusive.txt) > profiles/${TAG}_pcie_inclusive.txt
(echo "$HDR"; echo "# command: python tools/multi_bench.py   (jj_multi_* with page-locked host buffers on ONE GPU listed once and twice, beside the single-context entry points)"; grep -v amdgpu gpurun_out/${TAG}_multi_bench.txt) > profiles/${TAG}_multi_bench.txt
(echo "$HDR"; echo "# command: python tools/msm_dev_finish.py"; grep -v amdgpu gpurun_out/${TAG}_msm_dev_finish.txt) > profiles/${TAG}_msm_dev_finish.txt
(echo "$HDR"; grep -v amdgpu gpurun_out/${TAG}_fixedbase_select_pmc.txt) > profiles/${TAG}_fixedbase_select_pmc.txt
[ -s gpurun_out/${TAG}_stall_pmc.txt ] && (echo "$HDR"; echo "# command: bash tools/stall_pmc.sh"; grep -v amdgpu gpurun_out/${TAG}_stall_pmc.txt) > profiles/${TAG}_stall_pmc.txt
(echo "$HDR"; echo "# command: ./experiments/lds_probe/energy_probe ; ./experiments/lds_probe/probe"; cat gpurun_out/${TAG}_issue_energy_probe.txt) > profiles/${TAG}_issue_energy_probe.txt
(echo "$HDR"; echo "# command: python tests/host_tail_time.py 1 8 (default, then scalar: the scalar chain forced); then bench.py --workload msm --log2n 17 with the two chains in turn"; grep -v amdgpu gpurun_out/${TAG}_host_tail.txt) > profiles/${TAG}_host_tail.txt
(echo "$HDR"; echo "# command: python tests/soak_host.py 120"; grep -v amdgpu gpurun_out/${TAG}_soak_host.txt | tail -4) > profiles/${TAG}_soak_host.txt
(echo "$HDR"; echo "# command: python experiments/misc/vb_ct_window.py   (2^20 units: the table ladder, the constant-time ladder with signed 2-bit and 3-bit windows)"; grep -v amdgpu gpurun_out/${TAG}_vb_ct_window.txt) > profiles/${TAG}_vb_ct_window.txt
(echo "$HDR"; echo "# command: python experiments/misc/msm_reduce_l1_sweep.py 18 19 20 21 22"; grep -v amdgpu gpurun_out/${TAG}_msm_reduce_l1_sweep.txt) > profiles/${TAG}_msm_reduce_l1_sweep.txt
(echo "$HDR"; echo "# command: python experiments/misc/msm_sort_hist_ab.py"; grep -v amdgpu gpurun_out/${TAG}_msm_sort_hist_ab.txt) > profiles/${TAG}_msm_sort_hist_ab.txt
(echo "$HDR"; echo "# command: python experiments/misc/msm_allgather_pipeline.py 20 8   (the other seven ranks played by tools/loopback_comm.cpp)"; grep -v amdgpu gpurun_out/${TAG}_msm_allgather_pipeline.txt) > profiles/${TAG}_msm_allgather_pipeline.txt
[ -s gpurun_out/${TAG}_bench_msm20_rccl1_async4.json ] && cp gpurun_out/${TAG}_bench_msm20_rccl1_async4.json profiles/
(echo "$HDR"; grep -v amdgpu gpurun_out/${TAG}_peak_clock.txt) > profiles/${TAG}_peak_clock.txt
(echo "$HDR"; echo "# command: bench.py --opt <key>=<0|1> --workload msm --log2n 17 --steps 10 --warmup 3, alternating (msm_front1=0: the four-launch front end of round 5; msm_acc_lds=0: bucket offsets read from global memory)"; cat gpurun_out/${TAG}_msm17_ab.txt) > profiles/${TAG}_msm17_ab.txt
(echo "$HDR"; echo "# command: bench.py --opt msm_windows=<16|17> --workload msm --log2n <18|19> --steps 10 --warmup 3, alternating (16 = the planner's choice from 2^18 terms since round 6)"; cat gpurun_out/${TAG}_msm_windows_ab.txt) > profiles/${TAG}_msm_windows_ab.txt
(echo "$HDR"; echo "# command: python tests/soak_jobs.py 180 9000"; grep -v amdgpu gpurun_out/${TAG}_soak_jobs.txt | tail -4) > profiles/${TAG}_soak_jobs.txt
(echo "$HDR"; echo "# command: ./experiments/mad_banks/probe   (streams of 64 multiply-adds per trip with explicit register numbers; best of 5 launches)"; cat gpurun_out/${TAG}_mad_banks.txt) > profiles/${TAG}_mad_banks.txt
(echo "$HDR"; echo "# command: ./experiments/sync_latency/probe   (launch -> the host knows, for a kernel that spins 5 / 300 us; median of 300)"; cat gpurun_out/${TAG}_sync_latency.txt) > profiles/${TAG}_sync_latency.txt
(echo "$HDR"; echo "# command: python bench.py --workload msm --log2n <17|18|20> --msm-async <jobs in flight> --opt msm_lanes=<lanes> --no-cpu-baseline --no-extras   (default since round 6: three lanes)"; cat gpurun_out/${TAG}_msm_lanes.txt) > profiles/${TAG}_msm_lanes.txt
python3 tools/design_numbers.py $TAG
