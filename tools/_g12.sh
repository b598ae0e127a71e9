rm -f gpurun_out/ab_hide.txt
b() { # label lib args
  local lbl=$1; local lib=$2; shift; shift
  JJ_LIB_PATH=$lib timeout 600 python bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lbl: %.1f M/s kernel_ms %.3f frac %.3f' % (d['value']/1e6, r['kernel_ms'], r['frac']))" >> gpurun_out/ab_hide.txt
}
for i in 1 2 3; do
  b fb_old experiments/probe_lib/libjj_old.so --workload fixedbase
  b fb_new "" --workload fixedbase
  b vb_old experiments/probe_lib/libjj_old.so
  b vb_new ""
  b dec_old experiments/probe_lib/libjj_old.so --workload decompress
  b dec_new "" --workload decompress
done
b msm_old experiments/probe_lib/libjj_old.so --workload msm
b msm_new "" --workload msm
cat gpurun_out/ab_hide.txt
