# Is the MSM accumulate kernel waiting on its gathers?  Variants rebuilt on the box: the shipped kernel, all gathers redirected to 1024
# cache-hot entries (wrong results, timing only), two entries in flight instead of one, and both.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
one() {
  rm -rf gpurun_out/accmem
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/accmem -o m -- python $GRAFT_REPO_ROOT/bench.py --workload msm --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1)
  python3 - <<'PY'
import glob, sqlite3
db = sqlite3.connect(glob.glob('gpurun_out/accmem/**/*results.db', recursive=True)[0])
for r in db.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%accumulate_seg%' group by name"):
    print("   %-40s calls %4d  avg %8.1f us  min %8.1f us" % (r[0][:40], r[1], r[2] / 1e3, r[3] / 1e3))
PY
  rm -rf gpurun_out/accmem
}
for flags in "" "-DJJ_ACC_IDXMASK=1023u" "-DJJ_ACC_DEPTH=2" "-DJJ_ACC_DEPTH=2 -DJJ_ACC_IDXMASK=1023u" ""; do
  JJ_CXXFLAGS="-DJJ_EXPERIMENTS $flags" python -m jubjub_amd.build --force > /dev/null 2>&1
  echo "== flags: '$flags'"; one
done
