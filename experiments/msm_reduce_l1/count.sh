#!/bin/bash
# instruction counts of the lane-form first level (compile only): bash experiments/msm_reduce_l1/count.sh
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I jubjub_amd/csrc -Rpass-analysis=kernel-resource-usage --cuda-device-only -S -o /tmp/l1.s experiments/msm_reduce_l1/probe.hip 2>&1 | grep -A10 "Function Name: k_reduce_l1" | grep -E "VGPRs:|Scratch|Occupancy"
python3 - <<'PY'
import re
s = open('/tmp/l1.s').read()
i = s.find('k_reduce_l1:'); j = s.find('s_endpgm', i)
body = s[i:j].split('\n')
ins = lambda lines: [l.strip() for l in lines if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
labels = [(k, l) for k, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)]
for a, (st, name) in enumerate(labels):
    en = labels[a + 1][0] if a + 1 < len(labels) else len(body)
    blk = ins(body[st:en])
    if blk:
        print("%-12s %5d instructions, %5d multiply-adds, %2d loads%s" % (name.split(':')[0], len(blk), sum(x.startswith(('v_mad_i64', 'v_mad_u64')) for x in blk),
              sum(x.startswith('global_load') for x in blk), "   <- the loop: two whole-lane additions" if 'Loop' in name else ""))
PY
