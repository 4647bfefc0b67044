#!/usr/bin/env python3
"""Print the SHA-256 of the GPU code embedded in a built librtmi.so (its .hip_fatbin section).

Two builds with the same id run the same kernels: a change that leaves the id alone is host-only and
cannot alter what the GPU executes.  Usage: python tools/device_code_id.py [path/to/librtmi.so]"""
import hashlib
import os
import subprocess
import sys
import tempfile

OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def device_code_id(so_path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fatbin.bin")
        subprocess.check_call([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", so_path, out])
        with open(out, "rb") as f:
            data = f.read()
    if not data:
        raise SystemExit(f"{so_path}: no .hip_fatbin section")
    return hashlib.sha256(data).hexdigest(), len(data)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "raytracer_amd", "librtmi.so")
    h, n = device_code_id(path)
    print(f"{h}  {n} bytes  {os.path.relpath(path)}")
