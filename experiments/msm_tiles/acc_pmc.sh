cd /root/repo; export TMPDIR=/tmp
for m in segments tiles; do
  D=/tmp/pmc_$m; rm -rf $D
  (cd /tmp && JJ_MSM_ACCUM=$m timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D -o pmc -- python /root/repo/bench.py --workload msm --log2n 20 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $D.log 2>&1)
  f=$(find $D -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$m" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "accumulate" in n:
        acc[n.split("(")[0]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in acc.items():
    print(sys.argv[2], k, {c: "%.4g" % (sum(x[0] for x in l) / len(l)) for c, l in v.items()}, "avg ns %.0f" % (sum(x[1] for l in v.values() for x in l) / sum(len(l) for l in v.values())))
PY
done
