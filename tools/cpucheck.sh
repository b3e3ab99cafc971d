cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; grep Cpus_allowed_list /proc/self/status; python3 -c 'import os; print("affinity", len(os.sched_getaffinity(0)))'
python3 - <<'PY'
import subprocess, sys, time
code = "import time\nt=time.time()\nn=0\nwhile time.time()-t<2.0:\n    for i in range(100000): n+=i\nprint(n//100000)"
for N in (1, 8, 32, 64, 128, 256):
    ps=[subprocess.Popen([sys.executable,'-c',code],stdout=subprocess.PIPE,text=True) for _ in range(N)]
    tot=sum(int(p.stdout.read()) for p in ps)
    print(N, tot, tot/N)
PY
