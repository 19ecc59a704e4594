#!/usr/bin/env python3
"""Runs a selection of the GPU tests N times, every run in a FRESH process, and counts the failing runs: how the early-completion
of kernel-bound events was found - 40 repetitions inside one warm process never failed, 16 fresh processes
failed 5-7 times.   repeat_tests.py N [pytest -k expression]   e.g.  repeat_tests.py 16 failure_paths"""
import subprocess
import sys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
expr = sys.argv[2:] and ["-k", " ".join(sys.argv[2:])] or []
fails = 0
for r in range(n):
    p = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "-p", "no:cacheprovider"] + expr, capture_output=True, text=True)
    tail = [l for l in p.stdout.splitlines() if l.startswith("FAILED") or " passed" in l or " failed" in l]
    print("run", r, "rc", p.returncode, " | ".join(tail[-3:]), flush=True)
    fails += p.returncode != 0
print("failing runs:", fails, "of", n, flush=True)
