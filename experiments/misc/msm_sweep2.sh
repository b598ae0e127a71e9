# MSM tuning sweeps of round 3 (run on the GPU box): window count W, small-batch threshold, chunk sizes
cd $GRAFT_REPO_ROOT
S=experiments/misc/msm_sweep.py
for l in 10 12 13 14 15; do python $S $l JJ_MSM_SMALL_MAX=0,1000000 JJ_MSM_WINDOWS=-; done
for l in 12 13 14; do python $S $l JJ_MSM_SMALL_MAX=0 JJ_MSM_WINDOWS=22,24,26,28,32; done
for l in 15 16 17; do python $S $l JJ_MSM_SMALL_MAX=0 JJ_MSM_WINDOWS=18,19,20,21,22,23; done
for l in 18 19 20; do python $S $l JJ_MSM_WINDOWS=16,17,18,19; done
python $S 17 JJ_MSM_WINDOWS=19,20,21 JJ_MSM_CHUNK=8,16,32 JJ_MSM_REDUCE_CHUNK=4,8,16
