import sys, time, json, os
sys.path.insert(0, "/root/repo")
import bench
for threads, B in ((128, 8), (64, 8), (32, 8)):
    t0 = time.time()
    r = bench._cpu_run(threads, 200, B, {"base": ["train"]}, warm=1, timed=2)
    print(threads, B, r, round(time.time() - t0, 1), flush=True)
