"""The side kernels of a pipelined batch must FIT beside a resident scan workgroup: a CU that holds one
k_scan workgroup (12 waves, three per SIMD) has 512 - 3 x (the scan's registers, in granules of 8) vector registers per SIMD
and 160 KiB - the scan's LDS left, the LDS possibly in two pieces.  Nothing in the compiler can be told "at most 56
registers" (amdgpu_num_vgpr is ignored, amdgpu_waves_per_eu stops at 64), so the limits are checked on the BUILT library:
this test reads the register / LDS figures out of the gfx950 code objects inside libanorag_hip.so."""
import os

import pytest

from codeobj import kernel_metadata

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "..", "ano-rag_amd", "anorag_hip", "libanorag_hip.so")

SIMD_REGS = 512
LDS_BYTES = 160 * 1024
SHADOW_LDS_CAP = 29 * 1024  # = csrc kShadowLds: half of what the scan leaves, so that either piece holds it


def _alloc(k):
    v = k[".vgpr_count"]  # (already includes the accumulator registers on gfx90a+)
    return (v + 7) // 8 * 8


@pytest.fixture(scope="module")
def md():
    if not os.path.exists(SO):
        pytest.skip("libanorag_hip.so not built")
    return kernel_metadata(SO)


def _pick(md, *needles):
    out = {n: k for n, k in md.items() if all(t in n for t in needles)}
    assert out, f"no kernel matching {needles}"
    return out


def test_the_scan_leaves_56_registers_per_simd(md):
    for name, k in _pick(md, "k_scanILb0ELi8ELi768").items():  # the main scan at dims that are multiples of 256
        assert k[".max_flat_workgroup_size"] == 768
        assert SIMD_REGS - 3 * _alloc(k) >= 56, (name, k[".vgpr_count"])
        assert k[".sgpr_count"] <= 104


@pytest.mark.parametrize("needle", ["k_prepq", "k_sample", "k_select_shadow", "k_rescore_shadow", "k_finalize"])
def test_shadow_kernels_fit_beside_a_scan_workgroup(md, needle):
    scan = max(_alloc(k) for k in _pick(md, "k_scanILb0ELi8ELi768").values())
    free_regs = SIMD_REGS - 3 * scan
    for name, k in _pick(md, needle).items():
        assert k[".max_flat_workgroup_size"] == 256, name            # four waves: one per SIMD
        assert _alloc(k) <= free_regs, (name, k[".vgpr_count"], free_regs)
        assert k[".group_segment_fixed_size"] <= SHADOW_LDS_CAP, name  # static LDS; the dynamic part is bounded in index.hip
        assert k.get(".private_segment_fixed_size", 0) == 0, name      # no scratch: a spill would sit in the scan's HBM stream


def test_the_twelve_bit_scan_keeps_three_waves_per_simd_without_scratch(md):
    """k_scan<.., F12 = true> (ANR_OPT_SCAN_BITS 12: the image the 10 M-row headline streams) unpacks its operand in
    registers; it must still fit three waves per SIMD (<= 168 registers) and spill nothing"""
    picked = {n: k for n, k in _pick(md, "k_scanILb0ELi8ELi768").items() if n.endswith("ELb1EEEvNS_10ScanParamsE")}
    assert len(picked) == 2          # streaming (non-temporal) and cache-resident loads
    for name, k in picked.items():
        assert _alloc(k) <= 168, (name, k[".vgpr_count"])
        assert k.get(".private_segment_fixed_size", 0) == 0 and k.get(".vgpr_spill_count", 0) == 0, name
