# per-kernel breakdown of one MSM on the GPU box:  bash experiments/misc/msm_breakdown.sh [log2n] [env assignments...]
cd $GRAFT_REPO_ROOT
LOG2N=${1:-20}; shift
for kv in "$@"; do export "$kv"; done
timeout 600 python bench.py --workload msm --log2n $LOG2N --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['verified'], d['roofline']['frac'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "msm" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/msm2p
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/msm2p -o m -- python $GRAFT_REPO_ROOT/bench.py --workload msm --log2n $LOG2N --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import glob, sqlite3
db = sqlite3.connect(glob.glob('gpurun_out/msm2p/**/*results.db', recursive=True)[0])
tot = 0
for r in db.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%msm%' or name like '%seg%' or name like '%scan%' or name like '%sum_groups%' or name like '%soa_to%' group by name order by sum(duration) desc limit 24"):
    per = r[1] / 96.0
    tot += r[2] * per
    print("%-60s %5d %10.0f %10d" % (r[0][:60], r[1], r[2], r[3]))
print("sum of kernel time per MSM: %.1f us" % (tot / 1e3))
PY
rm -rf gpurun_out/msm2p
