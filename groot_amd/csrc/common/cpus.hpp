// cpus.hpp -- CPUs this process may really use: min(affinity mask, cgroup CPU quota); GROOT_THREADS overrides.  Containers show
// every hardware thread of the box (std::thread::hardware_concurrency) while granting a fraction of them: oversubscribing that
// quota only adds context switches.  Shared by the host library and the device library (its open builds tables on host threads).
#pragma once

#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace groot {

inline unsigned granted_cpus()
{
    static const unsigned cached = []() -> unsigned {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)std::max(1, CPU_COUNT(&set));
        // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us
        double quota = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64];
            long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / (double)period;
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            long long qv = -1, period = 0;
            if (fscanf(g, "%lld", &qv) != 1) qv = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 0; fclose(h); }
            if (qv > 0 && period > 0) quota = (double)qv / (double)period;
        }
        if (quota >= 1.0) n = std::min<unsigned>(n, (unsigned)(quota + 0.5));
        if (const char *e = getenv("GROOT_THREADS")) { const int v = atoi(e); if (v > 0) n = (unsigned)v; }
        return std::max(1u, n);
    }();
    return cached;
}

} // namespace groot
