# fixed-base LDS kernel with 512 (2 waves per SIMD, 203 VGPRs) or 768 threads per workgroup (3 waves per SIMD, 168 VGPRs + 128 B scratch), rebuilt on the box
cd $GRAFT_REPO_ROOT
one() { timeout 600 python bench.py --workload fixedbase --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench', round(d['value']/1e6,1), 'M/s kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4), d['verified'])"; }
for t in 512 768 640 512 768; do
  JJ_CXXFLAGS="-DJJ_EXPERIMENTS -DJJ_FB_THREADS=$t" python -m jubjub_amd.build --force > /dev/null 2>&1
  echo "== JJ_FB_THREADS=$t"; one
done
