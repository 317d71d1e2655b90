#!/usr/bin/env python3
"""rocFFT plan-build time of a fresh process (GPU box): children are started one after another with
different ROCFFT_RTC_CACHE_PATH settings, each builds the plan for (T, batch) once and reports the
host time of the build and of the first execution.
    python tools/fft_plan_timing.py [T] [K]"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
t_imp = time.perf_counter()
import numpy as np
from psa_amd import _hip
T, K = int(sys.argv[1]), int(sys.argv[2])
t0 = time.perf_counter()
eng = _hip.Engine(0)
t1 = time.perf_counter()
rng = np.random.default_rng(0)
vel = rng.standard_normal((T, 64, 3)).astype(np.float32)
eng.ensure_resident(0, vel)
eng.oneoff_stats()
kv = np.linspace(0.1, 1, K, dtype=np.float32)[:, None] * np.ones(3, np.float32)
t2 = time.perf_counter()
eng.project(0, np.zeros((64, 3), np.float32), kv)
eng.synchronize()
t3 = time.perf_counter()
one = eng.oneoff_stats()
eng.project(0, np.zeros((64, 3), np.float32), kv)
eng.synchronize()
t4 = time.perf_counter()
print(json.dumps(dict(engine_create_ms=1e3 * (t1 - t0), first_project_ms=1e3 * (t3 - t2), plan_ms=one["rocfft_plan"],
                      second_project_ms=1e3 * (t4 - t3))))
''' % str(ROOT)


def child(env_extra, T, K):
    env = dict(os.environ, **env_extra)
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, "-c", CHILD, str(T), str(K)], env=env, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        d = {"error": out.stderr[-400:]}
    d["process_wall_s"] = round(wall, 2)
    return d


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    cache = "/tmp/psa_rocfft_rtc_test.db"
    if os.path.exists(cache):
        os.remove(cache)
    print("HOME", os.environ.get("HOME"), "XDG_CACHE_HOME", os.environ.get("XDG_CACHE_HOME"), flush=True)
    for label, env in [("default env, process 1", {}), ("default env, process 2", {}),
                       ("ROCFFT_RTC_CACHE_PATH, process 1", {"ROCFFT_RTC_CACHE_PATH": cache}),
                       ("ROCFFT_RTC_CACHE_PATH, process 2", {"ROCFFT_RTC_CACHE_PATH": cache}),
                       ("ROCFFT_RTC_CACHE_PATH, process 3, other K", {"ROCFFT_RTC_CACHE_PATH": cache, "_K": "1"})]:
        k = 37 if env.pop("_K", None) else K
        print(label, json.dumps(child(env, T, k)), flush=True)
    print("cache file bytes", os.path.getsize(cache) if os.path.exists(cache) else None)
    for p in (Path.home() / ".cache" / "rocFFT", Path(os.environ.get("XDG_CACHE_HOME", "/nonexistent")) / "rocFFT"):
        print(p, sorted(x.name for x in p.iterdir()) if p.exists() else "absent")
