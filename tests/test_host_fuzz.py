"""Host code under AddressSanitizer + UBSan: OBJ/MTL loader and image writers fed with malformed and random
input (tests/cpp/host_fuzz.cpp).  No GPU code is involved."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_loader_and_writers_survive_random_input(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), "host_fuzz"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([os.path.join(HERE, "cpp", "host_fuzz"), str(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "host fuzz ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
