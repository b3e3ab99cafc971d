python -m pytest tests -m gpu -x -q 2>&1 | tail -6
run() { w=$1; shift
  env "$@" python tools/kernel_path_probe.py $w 6 ${READS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d['stage_ms']
print('$*', d['workload'], d['reads'], round(d['value'],1), {k: round(m[k],2) for k in ('sketch_seed','first_seed_kernel','list_pass','schedule','align','sort','wall')}, d['counts']['walked_reads'], flush=True)"
}
for w in sub1 mixed99 c2_nomemo headline; do run $w A=1; done
READS=2000000 run mixed99 A=1
