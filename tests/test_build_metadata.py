"""Resources of the headline kernels as BUILT (code-object metadata of libquest_hip.so; no GPU): no scratch, no spills,
static LDS small enough for two 8-wave workgroups per CU next to the dynamic staging arrays.  Guards against the kind
of regression round 4 hit while refactoring: a by-reference hand-over of the kernel parameters left a 32-byte slice of
them in memory, the compiler "promoted" it to LDS (16 KiB per workgroup), one workgroup fit a CU instead of two and the
headline launch took 24 us instead of 12 -- with every parity test green."""
import pytest

from quest_amd.build import LIB, kernel_metadata


@pytest.fixture(scope="module")
def meta():
    return kernel_metadata(LIB)


def _one(meta, fragment):
    hits = [v for k, v in meta.items() if fragment in k]
    assert len(hits) == 1, (fragment, [k for k in meta if fragment in k])
    return hits[0]


@pytest.mark.parametrize("fragment,max_lds", [
    ("sparse_decode_kernelILi128ELi16ELi8ELi8ELi4E", 14 * 1024),   # cfg 3: column-range ownership, first generation
    ("sparse_decode_kernelILi128ELi16ELi8ELi8ELi3E", 14 * 1024),   # the slot-ownership twin
    ("sparse_decode_kernelILi128ELi16ELi8ELi8ELi1E", 14 * 1024),   # batched launches (cfg 3 x 8, cfg 5)
    ("sparse_decode_kernelILi128ELi16ELi24ELi8ELi5E", 17 * 1024),  # cfg 4: column-range ownership, long rows
    ("sparse_decode_kernelILi128ELi16ELi24ELi8ELi2E", 18 * 1024),
    ("sparse_decode_kernelILi128ELi16ELi0ELi4ELin1E", 3 * 1024),   # index-list launches (reference op sequence)
    ("sparse_decode_kernelILi128ELi16ELi0ELi8ELin1E", 5 * 1024),
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


def test_streaming_kernels_have_no_scratch(meta):
    for frag in ("estimate_kernelILi128ELi1E", "estimate_kernelILi128ELi4E", "merge_states_kernelILi128E",
                 "shared_decode_kernelILi128ELi1E", "shared_decode_kernelILi128ELi4E", "append_decode_kernel"):
        hits = [v for k, v in meta.items() if frag in k]
        assert hits, frag
        for v in hits:
            assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (frag, v)
