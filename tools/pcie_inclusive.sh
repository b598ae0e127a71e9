#!/bin/bash
# Runs ON THE GPU BOX: the host-pointer (drop-in) path of the C ABI priced against the host link.
#   gpurun -- 'bash tools/pcie_inclusive.sh r4'  ->  gpurun_out/<tag>_pcie_probe.txt, <tag>_pcie_inclusive.txt, <tag>_bench_host_*.json
TAG=${1:-r4}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
[ -x tools/pcie_probe ] || /opt/rocm/bin/hipcc -O2 -o tools/pcie_probe tools/pcie_probe.cpp -lpthread
./tools/pcie_probe > gpurun_out/${TAG}_pcie_probe.txt 2>&1
OUT=gpurun_out/${TAG}_pcie_inclusive.txt
: > $OUT
line() {  # name, bench args...
  local name=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --passes $( [[ "$*" == *msm* ]] && echo 8 || echo 4 ) --no-cpu-baseline > gpurun_out/${TAG}_bench_host_$name.json 2>/dev/null
  python - "$name" gpurun_out/${TAG}_bench_host_$name.json >> $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); p = d["roofline"]["pcie"]
extra = ""
if p.get("caller_alloc_free_ms_per_pass") is not None:
    extra = "  [inside the call: %.2f ms = %.1f M units/s; caller's free + alloc of the result array: %.2f ms per pass]" % (
        p["call_ms_per_pass"], p["units_per_s_inside_the_call"] / 1e6, p["caller_alloc_free_ms_per_pass"])
print("%-36s %7.1f M units/s  (device-resident %6.1f M/s, ratio %.3f)  %6.2f ms/pass  H2D %5.1f GB/s  D2H %5.1f GB/s  link frac %.3f  bound by %-7s  verified %s / block %s / second pass %s%s" % (
    sys.argv[1], d["value"] / 1e6, d["device_resident"]["value"] / 1e6, d["host_over_device_resident"], p["ms_per_pass"], p["h2d_GBps"], p["d2h_GBps"], p["frac"],
    p["bound_by"], d.get("verified"), d.get("verified_block", {}).get("ok"), d.get("all_units_equal_second_pass"), extra))
PY
}
for hb in pinned pageable; do
  line varbase_$hb --workload varbase --host-buffers $hb
  line fixedbase_$hb --workload fixedbase --host-buffers $hb
  line fixedbase_compressed_$hb --workload fixedbase --host-buffers $hb --compressed
  line decompress_$hb --workload decompress --host-buffers $hb
done
# pageable arrays page-locked in place for the call (round 3's way) instead of the bounce path through the context's staging buffers
line fixedbase_pageable_register --opt pipe_pageable_register=1 --workload fixedbase --host-buffers pageable
line decompress_pageable_register --opt pipe_pageable_register=1 --workload decompress --host-buffers pageable
# a caller that allocates a NEW result array per call (vec![0u8; 64 * n]): bounce (default) and in-place page-locking with / without the pre-fault
line fixedbase_fresh --workload fixedbase --host-buffers fresh
line fixedbase_fresh_register --opt pipe_pageable_register=1 --workload fixedbase --host-buffers fresh
line fixedbase_fresh_register_noprefault --opt pipe_pageable_register=1 --opt pipe_prefault=0 --workload fixedbase --host-buffers fresh
line decompress_fresh --workload decompress --host-buffers fresh
line varbase_fresh --workload varbase --host-buffers fresh
# ... and the same caller taking its result buffers from the library's pool (jj_result_acquire / _release, round 5): a DIFFERENT page-locked buffer per call
line fixedbase_pooled --workload fixedbase --host-buffers pooled
line decompress_pooled --workload decompress --host-buffers pooled
line varbase_pooled --workload varbase --host-buffers pooled
# MSM of host arrays (96 bytes per term in, 64 bytes out): two to eight passes whose copies run beside the previous pass's kernels, and one pass
# after the whole copy (round 3)
for hb in pinned pageable; do
  line msm20_$hb --workload msm --host-buffers $hb
  line msm22_$hb --workload msm --log2n 22 --host-buffers $hb
done
line msm20_pinned_one_pass --opt msm_host_split=0 --workload msm --host-buffers pinned
line msm22_pinned_one_pass --opt msm_host_split=0 --workload msm --log2n 22 --host-buffers pinned
# uniform chunks (no short first / last chunk)
line fixedbase_pinned_uniform_chunks --opt pipe_ramp=0 --workload fixedbase --host-buffers pinned
line decompress_pinned_uniform_chunks --opt pipe_ramp=0 --workload decompress --host-buffers pinned
cat $OUT
