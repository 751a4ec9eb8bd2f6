for w in 0 2 1; do DVQ_BENCH_LANES=0 DVQ_DECODE_WAVE_ATTN=$w timeout 300 python bench_extra.py --workload sampling --bs 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('wave_attn=$w', {k:v['token_steps_per_sec'] for k,v in d['by_batch'].items()})"; done
