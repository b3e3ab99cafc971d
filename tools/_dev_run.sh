run() { w=$1; shift
  env "$@" python tools/kernel_path_probe.py $w 6 ${READS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d['stage_ms']
print('$*', d['workload'], d['reads'], round(d['value'],1), {k: round(m[k],2) for k in ('sketch_seed','first_seed_kernel','list_pass','schedule','align','sort','wall')}, d['counts']['walked_reads'], flush=True)"
}
for v in lw4_1 lw4_2 lw4_4 lw3_4; do
 for w in mixed99 mixed90 sub1; do run $w GROOT_HIP_LIB=build/$v/libgroot_hip.so; done
done
run mixed90 A=1
