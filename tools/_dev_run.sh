python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for pc in 4 2 1; do
  for w in sub1 mixed99 c2_nomemo; do
    GROOT_DEV_ALIGN_PER_CU=$pc python tools/kernel_path_probe.py $w 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d['stage_ms']
print('per_cu=$pc', d['workload'], round(d['value'],1), {k: round(m[k],2) for k in ('sketch_seed','first_seed_kernel','list_pass','schedule','align','sort','wall')}, d['counts']['walked_reads'])"
  done
done
