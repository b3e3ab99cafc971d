GROOT_BAM_STATS=1 python bench.py --steps 5 --no-legs --no-cpu --no-host-fed 2> /tmp/e.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['cli_e2e']; print(round(c['value'],2), round(c['stream_value'],2), c['phases_s'])"
grep "groot bam" /tmp/e.log | tail -4
df -h /tmp | tail -1; mount | grep -E " /tmp | / " | head -3
python - <<'PY'
import os, time
b = os.urandom(1<<26) * 16   # 1 GiB
t=time.time(); 
with open('/tmp/x.bin','wb') as f: f.write(b)
print('write 1GiB to /tmp: %.2f s' % (time.time()-t)); os.remove('/tmp/x.bin')
t=time.time();
with open('/dev/shm/x.bin','wb') as f: f.write(b)
print('write 1GiB to /dev/shm: %.2f s' % (time.time()-t)); os.remove('/dev/shm/x.bin')
PY
