#!/usr/bin/env python3
"""Build examples/pybind_binding.cpp (the reference-side PyBind binding of INTEGRATION.md section B) and, on a
GPU box, run its parity test (tests/test_gpu_cpp_binding.py: same bits as quest_amd._kernels for every op).

    python scripts/check_cpp_binding.py --build-only     (no GPU needed)
    python scripts/check_cpp_binding.py                  (on the GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import build_binding  # noqa: E402

print("built:", build_binding.build())
if "--build-only" not in sys.argv:
    sys.exit(subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_cpp_binding.py"), "-q",
                              "-m", "gpu"]))
