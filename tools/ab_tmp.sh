set -u
cd "$GRAFT_REPO_ROOT"
for w in mixed99 mixed90 c2_nomemo sub1_nomemo sub1 headline; do
  echo "== $w"
  timeout 300 python tools/kernel_path_probe.py $w 16 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"list_pass": [0-9.]*\|"alignments": [0-9]*'
done
echo "== 2M"; READS=2000000 timeout 300 python tools/mixed_leg_probe.py 0.99 256 24 2>&1 | tail -1 | grep -o "'Mreads_s': [0-9.]*"
