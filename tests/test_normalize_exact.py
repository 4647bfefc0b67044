"""rtx::normalize shares one division among the three quotients (raytracer_amd/csrc/rtmi_exact.h); the reference
divides three times (Utilities.swift:27-30).  The two must agree bit for bit: tests/cpp/normalize_check.cpp draws
vectors from families that stress the rounding of a quotient and compares."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shared_division_normalize_equals_three_divisions():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), "normalize_check"])
    r = subprocess.run([os.path.join(HERE, "cpp", "normalize_check"), "60", "2026"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
    through = int(r.stdout.split(" vectors, ")[1].split()[0])
    assert through > 40_000_000  # most vectors do take the shared division
