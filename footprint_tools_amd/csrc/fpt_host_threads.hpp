// fpt_host_threads.hpp -- how many threads a host-side team may usefully have: the hardware's
// count, cut to the process's affinity mask and to the CPU quota of its control group.  (A
// container on a 256-thread machine with a quota of 16 CPUs reports 256 from
// std::thread::hardware_concurrency; a team of 256 then spends its time being throttled --
// measured on the GPU box: 458 MB of track text deflated in 1.2 s on 256 threads.)
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <thread>

inline int fpt_host_cpus() {
    static const int n = []() {
        int c = (int)std::thread::hardware_concurrency();
        if (c <= 0) c = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a > 0) c = std::min(c, a);
        }
        long long quota = -1, period = -1;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[32];
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') sscanf(q, "%lld", &quota);
            fclose(f);
        } else {  // cgroup v1
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1) quota = -1;
                fclose(g);
            }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%lld", &period) != 1) period = -1;
                fclose(g);
            }
        }
        if (quota > 0 && period > 0) c = std::min<long long>(c, std::max<long long>(1, (quota + period - 1) / period));
        return c;
    }();
    return n;
}
