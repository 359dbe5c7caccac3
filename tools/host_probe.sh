#!/bin/bash
# what the GPU box's host side really offers (cores visible vs. cores granted)
echo "nproc: $(nproc)  python cpu_count: $(python -c 'import os; print(os.cpu_count())')  affinity: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cpu.cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)|MHz" | head -8
free -g | head -2
python - <<'PY'
import os, time, zlib, threading
raw = os.urandom(1 << 16) * 4 + bytes(1 << 18) + (b"ACGTTGCA" * 8192)
comp = [zlib.compress(raw[i:i + 65280], 1) for i in range(0, len(raw), 65280)] * 1024
tot = sum(len(zlib.decompress(c)) for c in comp[:8]) / 8 * len(comp)
def work(lo, hi):
    for c in comp[lo:hi]:
        zlib.decompress(c)
for nt in (1, 4, 16, 32, 64, 128, 256):
    per = (len(comp) + nt - 1) // nt
    th = [threading.Thread(target=work, args=(i * per, min(len(comp), (i + 1) * per))) for i in range(nt)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("zlib.decompress in %3d python threads: %.3f s  %.2f GB/s" % (nt, dt, tot / dt / 1e9))
PY
