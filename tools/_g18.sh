timeout 900 python -m pytest tests/test_gpu_host_path.py -x -q 2>&1 | tail -3
rm -f gpurun_out/bounce.txt
run() { # label env... -- args
  local lbl=$1; shift
  env "$@" timeout 300 python bench.py --workload $WL --host-buffers $HB --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/bounce_$lbl.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lbl: %.1f M/s  ratio %.3f  ms/pass %.2f verified %s block %s' % (d['value']/1e6, d['host_over_device_resident'], d['roofline']['pcie']['ms_per_pass'], d.get('verified'), d.get('verified_block',{}).get('ok')))" >> gpurun_out/bounce.txt
}
for WL in fixedbase decompress varbase; do
  HB=fresh;    run ${WL}_fresh_bounce JJ_PIPE_PAGEABLE=bounce JJ_PIPE_DEBUG=1
  HB=fresh;    run ${WL}_fresh_register JJ_PIPE_PAGEABLE=register
  HB=pageable; run ${WL}_pageable_bounce JJ_PIPE_PAGEABLE=bounce
  HB=pageable; run ${WL}_pageable_register JJ_PIPE_PAGEABLE=register
  HB=pinned;   run ${WL}_pinned X=1
done
WL=fixedbase; HB=fresh; for t in 2 4 8 16; do run fixedbase_fresh_bounce_threads$t JJ_PIPE_COPY_THREADS=$t; done
cat gpurun_out/bounce.txt; tail -4 gpurun_out/bounce_fixedbase_fresh_bounce.err
