#!/bin/bash
# Everything a change to the kernels has to pass on the CPU before it is run on a GPU.
#   bash tools/preflight.sh [fuzz cases, default 300]
# 1. product build (hipcc, gfx950) and the id of its GPU code  2. emulated kernels under ASan/UBSan against the oracle
# 3. the same under ThreadSanitizer (races between work-items)   4. random scenes   5. branches no run has taken
set -euo pipefail
cd "$(dirname "$0")/.."
CASES=${1:-300}
make -s -C raytracer_amd/csrc
python tools/device_code_id.py
make -s -C oracle
python -m pytest tests/test_emulated_kernels.py -x -q
python tests/emu_fuzz.py --cases "$CASES" --seed "$(date +%s)" --jobs 6
python tests/emu_fuzz.py --tsan --cases "$((CASES / 5))" --seed "$(date +%s)" --jobs 6
make -s -C tests/emu coverage
rm -rf /tmp/rtmi_cov && mkdir -p /tmp/rtmi_cov
EMU_RENDER_BIN=$PWD/tests/emu/build_cov/emu_render EMU_COVERAGE_FILE=/tmp/rtmi_cov/run \
  python -m pytest tests/test_emulated_kernels.py -x -q -k "not selftest and not api and not races"
python tools/emu_coverage_report.py /tmp/rtmi_cov/run.* | head -60
echo "preflight passed: now the GPU tests (python -m pytest tests -m gpu), each step under its own timeout"
