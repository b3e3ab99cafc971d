#!/bin/bash
# A/B of library variants on one workload (run on the GPU box): per variant directory under build/ the serial (one batch at a time) kernel
# durations of the align stage's kernels and the pipelined rate.   tools/ab_probe.sh WORKLOAD dir1 dir2 ...   ("." = the product build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=$1; shift
for L in "$@"; do
  LIB=build/$L/libgroot_hip.so; [ "$L" = "." ] && LIB=build/libgroot_hip.so
  D=$L; [ "$L" = "." ] && D=product
  rm -rf /tmp/ab/$D
  GROOT_HIP_LIB=$LIB PROBE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab/$D -o t -- python tools/kernel_path_probe.py $WL 3 > /tmp/ab_$D.json 2>/dev/null
  echo "== $L (serial): $(python -c "import json;d=json.loads(open('/tmp/ab_$D.json').read().strip().splitlines()[-1]);print('lean_reads',d['counts'].get('lean_reads'),'walked',d['counts']['walked_reads'])")"
  grep -h "align_lean\|align_kernel\|sketch_sig" $(find /tmp/ab/$D -name "*kernel_stats.csv") | awk -F'","|",|,' '{printf "   %-60s calls %s avg %.3f ms\n", substr($1,1,60), $2, $4/1e6}'
  echo "   pipelined: $(GROOT_HIP_LIB=$LIB python tools/kernel_path_probe.py $WL 12 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['value'],1),'Mreads/s  align',round(d['stage_ms']['align'],3),'lean',round(d['stage_ms'].get('lean_pass',0),3))")"
done
