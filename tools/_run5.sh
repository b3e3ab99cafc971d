python -m pytest tests/test_signature_path.py tests/test_gpu_parity.py tests/test_memo_tables.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
for w in "mixed99 10" "mixed99 12 2000000" "mixed90 6" "sub1 6" "c2_nomemo 4"; do
    python tools/kernel_path_probe.py $w 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['stage_ms']
print(d['workload'], d['reads'], 'value %.0f  seed %.2f (first %.2f list %.2f) align %.2f sort %.2f' % (d['value'], m['sketch_seed'], m['first_seed_kernel'], m['list_pass'], m['align'], m['sort']))"
done
