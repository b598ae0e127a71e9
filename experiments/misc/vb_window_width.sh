#!/bin/bash
# does a 4-bit window (8 + 1 table entries per lane instead of 16 + 1) cut the var-base table traffic, and
# at what price?  Builds the library twice on the GPU box (hipcc is there) and reports throughput + FETCH_SIZE / WRITE_SIZE.
set -e
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for W in 5 4; do
  JJ_CXXFLAGS="-DJJ_EXPERIMENTS -DJJ_VB_W=$W" python -m jubjub_amd.build --force > /dev/null
  echo "== window width $W"
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   %.2f M scalar-muls/s, k_varbase %.3f ms, verified %s' % (d['value']/1e6, r['kernel_ms'], d['verified']))"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/vbw_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/vbw_$C -o pmc -- python bench.py --steps 1 --warmup 1 --passes 1 --no-extras --no-cpu-baseline --no-verify > /dev/null 2>&1
    python - "$C" <<'PY'
import csv, glob, sys
c = sys.argv[1]
rows = [r for f in glob.glob('/tmp/vbw_%s/**/*counter_collection.csv' % c, recursive=True) for r in csv.DictReader(open(f)) if 'k_varbase<' in r['Kernel_Name'] and r['Counter_Name'] == c]
last = max(rows, key=lambda r: int(r['Start_Timestamp']))
kb = float(last['Counter_Value'])
print('   %s = %.0f KB per launch -> %.0f B per unit%s' % (c, kb, kb * 1024 * (2 if c == 'FETCH_SIZE' else 1) / (1 << 20), ' (x2: gfx950 half-count)' if c == 'FETCH_SIZE' else ''))
PY
  done
done
python -m jubjub_amd.build --force > /dev/null
