#!/bin/bash
# Round-4 evidence in one gpurun call (run from the repo root on the GPU box): parity suite, the driver line, soak, the timeline build of the align
# kernel (python __graft_entry__.py wc 3 first), then tools/profile_r04.sh (kernel trace + PMC passes) -> gpurun_out/f_*.{txt,json}, r04_*; copy into profiles/
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/f_pytest.txt
python bench.py --steps 20 > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
python tools/soak.py 8 60000 > gpurun_out/f_soak.txt 2>&1
( export GROOT_HIP_LIB=build/wc3/libgroot_hip.so
  for w in "sub1 3" "mixed99 3" "mixed99 4 2000000" "c2_nomemo 3"; do echo "== kernel_path_probe.py $w (build/wc3)"; python tools/kernel_path_probe.py $w 2>&1 | grep -E "timeline" | tail -8 | cut -c1-900; done ) > gpurun_out/f_timeline.txt 2>&1
bash tools/profile_r04.sh > gpurun_out/f_profile.log 2>&1
