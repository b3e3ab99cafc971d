#!/bin/bash
# Round-6 rocprofv3 evidence (run on the GPU box via gpurun from the repo root): per workload of tools/kernel_path_probe.py
#   1. --kernel-trace                -> profiles/r06_kernel_stats.csv   (per workload: calls / total / average / min / max ns of every groot +
#                                       rocprim kernel of the measured loop; launches of groot_hip_open and of torch are cut by time)
#   2. --pmc passes (own runs, no trace domains): SQ issue counters, FETCH_SIZE, WRITE_SIZE, L2 hits
#                                    -> profiles/r06_pmc.json            ({workload: {kernel: per-launch counters, hbm_bytes_per_launch}})
#   PROF_TAG=name: write profiles/name_* instead (A/B probes); PROF_QUICK=1: kernel trace + SQ counters only; PROF_COMMIT=hash is recorded in the json
#   usage: tools/profile_r06.sh [workloads...]     default: headline c2_nomemo sub1_nomemo sub1 mixed99 mixed90
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
WL="${@:-headline c2_nomemo sub1_nomemo sub1 mixed99 mixed90}"
P=/tmp/prof6; rm -rf $P; mkdir -p $P profiles gpurun_out
for w in $WL; do
  rocprofv3 --kernel-trace --output-format csv -d $P/$w/trace -o t -- python tools/kernel_path_probe.py $w 4 > gpurun_out/prof6_$w.json 2> gpurun_out/prof6_$w.err
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $P/$w/sq -o c -- python tools/kernel_path_probe.py $w 2 > /dev/null 2> gpurun_out/prof6_${w}_sq.err
  [ -n "${PROF_QUICK:-}" ] && continue
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/$w/fetch -o c -- python tools/kernel_path_probe.py $w 2 > /dev/null 2>> gpurun_out/prof6_${w}_sq.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/$w/write -o c -- python tools/kernel_path_probe.py $w 2 > /dev/null 2>> gpurun_out/prof6_${w}_sq.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/$w/tcc -o c -- python tools/kernel_path_probe.py $w 2 > /dev/null 2>> gpurun_out/prof6_${w}_sq.err
done
python tools/profile_r06_post.py $P $WL
cp profiles/${PROF_TAG:-r06}_kernel_stats.csv profiles/${PROF_TAG:-r06}_pmc.json gpurun_out/ 2>/dev/null
