#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in 1 0; do echo "DVQ_USE_HIPBLASLT=$v"; DVQ_USE_HIPBLASLT=$v timeout 300 python bench_extra.py --workload stage2 --no-cpu-baseline --steps 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d.get('roofline',{}).get('frac'));
for k,v in sorted(d.get('kernel_families',{}).items(), key=lambda kv:-kv[1]['ms_per_step'])[:10]: print('   ',k,v)"; done
