#!/bin/bash
# MSM counting sort: how many terms per thread should be in flight per trip (loads -> LDS atomics -> scattered 4-byte stores)?
set -e
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for U in 4 8 16 2; do
  JJ_CXXFLAGS="-DJJ_EXPERIMENTS -DJJ_MSM_SORT_UNROLL=$U" python -m jubjub_amd.build --force > /dev/null 2>&1
  rm -rf /tmp/msu; rocprofv3 --kernel-trace --stats -d /tmp/msu -o m -- python bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /tmp/msu.log 2>&1
  python - "$U" <<'PY'
import glob, sqlite3, sys, json
db = sqlite3.connect(glob.glob('/tmp/msu/**/*results.db', recursive=True)[0])
rows = dict(db.execute("select name, avg(duration) from kernels group by name").fetchall())
g = lambda k: sum(v for n, v in rows.items() if k in n) / 1e3
line = [l for l in open('/tmp/msu.log', errors='replace') if l.startswith('{')][0]
d = json.loads(line)
print('unroll %2s: k_msm_scatter %.1f us, k_msm_hist %.1f us, whole MSM %.3f ms per pass' % (sys.argv[1], g('k_msm_scatter'), g('k_msm_hist'), d['config']['ms_per_pass']))
PY
done
python -m jubjub_amd.build --force > /dev/null 2>&1
