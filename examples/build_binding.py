"""Build / load examples/pybind_binding.cpp -- the reference-side PyBind binding of INTEGRATION.md section B --
against libquest_hip.so, in-tree (build/binding_example/, git-ignored, travels to the GPU box with the snapshot).

    python examples/build_binding.py          # build (no GPU needed: it is host C++ linked against the HIP library)
"""
from __future__ import annotations

import importlib.util
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = "quest_binding_example"
BUILD_DIR = os.path.join(ROOT, "build", "binding_example")
SO_PATH = os.path.join(BUILD_DIR, NAME + ".so")
SRC = os.path.join(ROOT, "examples", "pybind_binding.cpp")


def up_to_date() -> bool:
    deps = [SRC, os.path.join(ROOT, "include", "quest_hip.h")]
    return os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(d) for d in deps)


def build(verbose: bool = False) -> str:
    """Compile with torch.utils.cpp_extension (g++ + ninja); returns the .so path."""
    import torch  # noqa: F401
    from torch.utils.cpp_extension import load

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import quest_amd.build as qb

    qb.build()
    if up_to_date():
        return SO_PATH
    lib_dir = os.path.join(ROOT, "quest_amd")
    os.makedirs(BUILD_DIR, exist_ok=True)
    load(name=NAME, sources=[SRC], extra_include_paths=[os.path.join(ROOT, "include"), "/opt/rocm/include"],
         extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
         extra_ldflags=[f"-L{lib_dir}", "-lquest_hip", f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-lamdhip64"],
         build_directory=BUILD_DIR, verbose=verbose, is_python_module=False)
    return SO_PATH


def load_module(allow_build: bool = True):
    """Import the extension module; builds it first when it is missing or stale and a host compiler exists.
    Returns None when it cannot be had (no prebuilt .so and no compiler)."""
    import torch  # noqa: F401  (libtorch symbols must be loaded before the extension)

    if not up_to_date():
        if not allow_build or shutil.which("g++") is None or shutil.which("ninja") is None:
            if not os.path.exists(SO_PATH):
                return None
        else:
            build()
    spec = importlib.util.spec_from_file_location(NAME, SO_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
