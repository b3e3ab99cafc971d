#!/bin/bash
# A/B helper for kernel experiments (GPU box): bench each build/libgroot_hip_<tag>.so in place of the product library.
#   tools/try_libs.sh tagA tagB ...     -> one line per tag: Mreads/s, ms/step, stage ms
cd "$GRAFT_REPO_ROOT" || exit 1
cp build/libgroot_hip.so /tmp/keep.so
for t in base "$@"; do
  if [ "$t" != base ]; then cp build/libgroot_hip_$t.so build/libgroot_hip.so; else cp /tmp/keep.so build/libgroot_hip.so; fi
  echo -n "$t: "
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['config']['stage_ms'], d['config']['per_step_counts']['alignments'])"
done
cp /tmp/keep.so build/libgroot_hip.so
