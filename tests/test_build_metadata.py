"""Resources of the headline kernels as BUILT (code-object metadata of libquest_hip.so; no GPU): no scratch, no spills,
static LDS small enough for two 8-wave workgroups per CU next to the dynamic staging arrays.  Guards against the kind
of regression round 4 hit while refactoring: a by-reference hand-over of the kernel parameters left a 32-byte slice of
them in memory, the compiler "promoted" it to LDS (16 KiB per workgroup), one workgroup fit a CU instead of two and the
headline launch took 24 us instead of 12 -- with every parity test green."""
import os

import pytest

from quest_amd.build import LIB, kernel_metadata

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists(os.path.join(LLVM, "llvm-readelf")),
                                reason="needs the built library and the ROCm LLVM tools (llvm-objcopy / llvm-readelf)")


@pytest.fixture(scope="module")
def meta():
    return kernel_metadata(LIB)


def _one(meta, fragment):
    hits = [v for k, v in meta.items() if fragment in k]
    assert len(hits) == 1, (fragment, [k for k in meta if fragment in k])
    return hits[0]


@pytest.mark.parametrize("fragment,max_lds", [
    ("sparse_decode_kernelILi128ELi16ELi8ELi8ELi3E", 14 * 1024),   # cfg 3: the timed single-sequence launch
    ("sparse_decode_kernelILi128ELi16ELi8ELi8ELi1E", 14 * 1024),   # batched two-launch form (cfg 5)
    ("sparse_decode_kernelILi128ELi16ELi8ELi8ELi8E", 15 * 1024),   # cfg 4: tiles front end (round 6: the page list holds 256 ids, + 512 B)
    ("sparse_decode_kernelILi128ELi16ELi24ELi8ELi2E", 18 * 1024),  # long rows without tile maxima
    ("sparse_decode_kernelILi128ELi16ELi0ELi4ELin1E", 3 * 1024),   # index-list launches (reference op sequence)
    ("sparse_decode_kernelILi128ELi16ELi0ELi8ELin1E", 5 * 1024),
    ("layer_decode_kernelILi128ELi8ELi8E", 14 * 1024),             # the one-launch layer (8 x cfg 3; + 12.3 KiB dynamic at 2069 pages)
])
def test_attention_kernels_fit_two_workgroups_per_cu(meta, fragment, max_lds):
    k = _one(meta, fragment)
    assert k["scratch"] == 0 and k["vgpr_spill"] == 0, k
    assert k["vgpr"] <= 128, k  # 8 waves x 2 workgroups per CU = 4 waves per SIMD
    assert k["lds"] <= max_lds, k


def test_no_d128_attention_instantiation_spills(meta):
    bad = {k: v for k, v in meta.items() if "sparse_decode_kernelILi128E" in k and (v["vgpr_spill"] or v["scratch"])}
    assert not bad, bad
    bad = {k: v for k, v in meta.items() if "sparse_decode_kernelILi64E" in k and (v["vgpr_spill"] or v["scratch"])}
    assert not bad, bad
    # head_dim 256 (not a BASELINE shape; the reference supports it): 8-wave instantiations are built for 2 waves per SIMD
    bad = {k: v for k, v in meta.items() if "sparse_decode_kernelILi256E" in k and (v["vgpr_spill"] or v["scratch"])}
    assert not bad, bad


def test_prefill_kernel_fits_two_workgroups_per_cu(meta):
    """csrc/prefill.hip is designed around two 4-wave workgroups per CU for head_dim 64 / 128 (two waves of DIFFERENT
    workgroups per SIMD, whose MFMA and softmax phases overlap): at most 256 VGPRs, at most half of the CU's 160 KiB of
    LDS; head_dim 256 runs one wave per SIMD.  No instantiation spills."""
    hits = {k: v for k, v in meta.items() if "14prefill_kernelILi" in k}
    assert len(hits) == 6, list(hits)  # head_dim 64 / 128 / 256 x (16-token pages, the generic page walk)
    for k, v in hits.items():
        assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (k, v)
        assert v["lds"] <= 80 * 1024, (k, v)
        if "ILi256E" not in k:
            assert v["vgpr"] <= 256, (k, v)


def test_streaming_kernels_have_no_scratch(meta):
    for frag in ("estimate_kernelILi128ELi1E", "estimate_kernelILi128ELi4E", "merge_states_kernelILi128E",
                 "shared_decode_kernelILi128ELi1E", "shared_decode_kernelILi128ELi4E", "append_decode_kernel"):
        hits = [v for k, v in meta.items() if frag in k]
        assert hits, frag
        for v in hits:
            assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (frag, v)


def test_import_refuses_a_library_built_from_other_sources(tmp_path):
    """the loaded binary is tied to the sources.  A copy of the tree's sources with ONE byte appended
    to a kernel file must make both `verify_build` and a fresh `import quest_amd._lib` refuse the prebuilt library; the
    untouched tree loads, and `needs_build()` follows the same hash (not file times)."""
    import os
    import shutil
    import subprocess
    import sys

    from quest_amd import build
    from quest_amd._lib import verify_build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert build.library_hash(LIB) == build.source_hash() and not build.needs_build()
    assert build.library_hash(LIB).encode() in __import__("quest_amd._lib", fromlist=["lib"]).lib.quest_build_info()
    verify_build(LIB)  # the tree as it stands
    shutil.copytree(os.path.join(root, "quest_amd", "csrc"), tmp_path / "quest_amd" / "csrc")
    os.makedirs(tmp_path / "include")
    shutil.copy(os.path.join(root, "include", "quest_hip.h"), tmp_path / "include" / "quest_hip.h")
    assert build.source_hash(str(tmp_path)) == build.source_hash()
    with open(tmp_path / "quest_amd" / "csrc" / "decode_device.cuh", "ab") as f:
        f.write(b" ")
    assert build.source_hash(str(tmp_path)) != build.source_hash()
    with pytest.raises(ImportError, match="stale"):
        verify_build(LIB, str(tmp_path))
    env = dict(os.environ, QUEST_SRC_ROOT=str(tmp_path), PYTHONPATH=root)
    env.pop("QUEST_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", "import quest_amd._lib"], env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode != 0 and "stale" in r.stderr, r.stderr[-400:]
    # a newer file time alone is not a change (checked on a copy: the working tree is not touched)
    shutil.copy(os.path.join(root, "quest_amd", "csrc", "decode_device.cuh"), tmp_path / "quest_amd" / "csrc" / "decode_device.cuh")
    os.utime(tmp_path / "quest_amd" / "csrc" / "quest_common.cuh")
    assert build.source_hash(str(tmp_path)) == build.source_hash() and not build.needs_build()
    # a tree without its sources: an ImportError that says what to do, not a raw FileNotFoundError
    with pytest.raises(ImportError, match="QUEST_SRC_ROOT"):
        verify_build(LIB, str(tmp_path / "nowhere"))
