# accumulate kernel compiled for 1 / 3 / 4 / 5 resident blocks per CU (VGPR budget 256 / 168 / 128 / 96), rebuilt and timed on one box
cd $GRAFT_REPO_ROOT
one() { timeout 600 python bench.py --workload msm --log2n 20 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench', round(d['value']/1e6,1), round(d['ms_per_step']/32,4))"; }
echo "== MINBLOCKS=4"; one; one
JJ_CXXFLAGS="-DJJ_EXPERIMENTS -DJJ_MSM_ACC_MINBLOCKS=1" python -m jubjub_amd.build --force > /dev/null 2>&1
echo "== MINBLOCKS=1"; one; one
JJ_CXXFLAGS="-DJJ_EXPERIMENTS -DJJ_MSM_ACC_MINBLOCKS=3" python -m jubjub_amd.build --force > /dev/null 2>&1
echo "== MINBLOCKS=3"; one; one
JJ_CXXFLAGS="-DJJ_EXPERIMENTS -DJJ_MSM_ACC_MINBLOCKS=5" python -m jubjub_amd.build --force > /dev/null 2>&1
echo "== MINBLOCKS=5"; one; one
