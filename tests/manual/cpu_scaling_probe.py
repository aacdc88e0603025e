import multiprocessing as mp, time, zlib, os
def work(_):
    d = os.urandom(1 << 16) * 4
    t = time.perf_counter(); n = 0
    while time.perf_counter() - t < 2.0:
        zlib.compress(d, 6); n += 1
    return n
if __name__ == "__main__":
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try: print(f, open(f).read().strip())
        except Exception as e: print(f, "n/a")
    print("affinity", len(os.sched_getaffinity(0)))
    for p in (1, 4, 8, 16, 32, 64, 128):
        with mp.Pool(p) as pool:
            r = pool.map(work, range(p))
        print(p, "procs:", sum(r) / 2.0, "compress/s total,", sum(r) / 2.0 / p, "per proc", flush=True)
