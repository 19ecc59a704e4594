import sys, subprocess
ids = [l.strip() for l in open("tools/_gpu_ids.txt") if "::" in l]
a, b, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
extra = sys.argv[4:]
sel = ids[a - 1:b]
fails = 0
for r in range(reps):
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider"] + extra + sel, capture_output=True, text=True)
    tail = [l for l in p.stdout.splitlines() if l.startswith("FAILED") or "passed" in l or "failed" in l]
    print("rep", r, "rc", p.returncode, " | ".join(tail[-3:]), flush=True)
    fails += p.returncode != 0
print("range", a, b, "failures", fails, "of", reps, flush=True)
