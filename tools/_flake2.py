import sys, pytest
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for i in range(n):
    rc = pytest.main(["tests", "-m", "gpu", "-x", "-q", "-k", "failure_paths or nan_path or euroc_frame_size or reuses_the_previous", "-p", "no:cacheprovider"])
    if rc != 0:
        bad += 1
        print("FAILED iteration", i)
print("failures:", bad, "of", n)
