#!/usr/bin/env python3
"""What the host side of a GPU box offers the host-fed workers: cores this process may use, cache sizes, SIMD flags, and the memory
bandwidth n threads reach together (numpy copies release the GIL).  One JSON line."""
import json
import os
import subprocess
import threading
import time

import numpy as np


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:  # noqa: BLE001
        return f"<{e}>"


def copy_bw(n_threads, mb=256, reps=4):
    src = [np.ones(mb << 20, dtype=np.uint8) for _ in range(n_threads)]
    dst = [np.empty(mb << 20, dtype=np.uint8) for _ in range(n_threads)]

    def work(i):
        for _ in range(reps):
            np.copyto(dst[i], src[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return 2.0 * n_threads * reps * (mb << 20) / dt / 1e9  # read + written


res = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_max": sh("cat /sys/fs/cgroup/cpu.max"),
       "model": sh("grep -m1 'model name' /proc/cpuinfo"), "sockets": sh("lscpu | grep -E 'Socket|NUMA node\\(s\\)|Thread|Core' | tr -s ' ' | tr '\\n' ';'"),
       "flags": [f for f in ("avx2", "avx512f", "avx512bw", "avx512vbmi", "bmi2", "gfni") if f in sh("grep -m1 flags /proc/cpuinfo").split()],
       "mem_total_gb": round(int(sh("grep MemTotal /proc/meminfo").split()[1]) / 1e6, 1), "cgroup_mem_max": sh("cat /sys/fs/cgroup/memory.max"),
       "tmp": sh("df -h /tmp | tail -1"), "shm": sh("df -h /dev/shm | tail -1")}
res["copy_GBps_read_plus_write"] = {str(t): round(copy_bw(t), 1) for t in (1, 4, 8, 16, 32, 64) if t <= (os.cpu_count() or 1)}
print(json.dumps(res))
