"""Static guard against round 4's headline regression: taking the address of a by-value kernel argument (`&p0` handed to an out-of-line
routine) made EVERY wavefront of the decode line copy its 288 bytes of arguments to scratch -- 36 MiB of HBM writes per launch, -4 % --
and nothing but a counter run showed it.  The code object says it at build time: `.private_segment_fixed_size` of the kernel.  Checked
here on the object the Makefile built (no GPU needed): the matrix-core decode kernels use no scratch at all, the dot2 kernels only what
their known register spills need, and nothing spills scalar registers inside the tile loop's budget."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "atoma-infer_amd", "build", "paged_decode.o")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_notes(obj, tmp):
    fb, co = os.path.join(tmp, "fb"), os.path.join(tmp, "co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co])
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out = {}
    for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1))
        out[re.search(r"\.name:\s+(\S+)", e).group(1)] = dict(scratch=g("private_segment_fixed_size"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"), vgpr=g("vgpr_count"))
    return out


@pytest.mark.skipif(not (os.path.exists(OBJ) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))), reason="needs the built object and the ROCm llvm tools")
def test_decode_kernels_use_no_scratch_for_their_arguments(tmp_path):
    ks = kernel_notes(OBJ, str(tmp_path))
    mqk = {n: v for n, v in ks.items() if "paged_decode_mqk_kernel" in n}
    assert len(mqk) >= 48
    for n, v in mqk.items():
        assert v["scratch"] == 0 and v["vspill"] == 0, f"{n}: {v} -- scratch in a matrix-core decode kernel (a kernel argument whose address escapes?)"
        assert v["vgpr"] <= 256 and v["sspill"] <= 4, (n, v)
    # the headline kernel itself: bf16, 4 q heads per wavefront, 3 tiles in flight, non-temporal, balanced line
    head = [v for n, v in mqk.items() if "INS_6bf16_tELi4ELi3ELb1ELb1ELb0E" in n]
    assert len(head) == 1 and head[0]["scratch"] == 0
    # dot2 kernels: scratch only where registers spill (the P = 4 variants: known and never dispatched by default); never the 288-byte argument copy
    for n, v in ks.items():
        if "paged_decode_kernel" in n and v["vspill"] == 0:
            assert v["scratch"] == 0, f"{n}: {v} -- scratch without a register spill"


GEN = os.path.join(ROOT, "atoma-infer_amd", "build", "attn_generic.o")


@pytest.mark.skipif(not (os.path.exists(GEN) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))), reason="needs the built object and the ROCm llvm tools")
def test_generic_attention_kernels_keep_their_state_in_registers(tmp_path):
    """attn_generic.hip's round-5 kernels: the 64-row prefill kernel is built for two wavefronts per SIMD (<= 256 registers, nothing in scratch --
    its first version with a second register stage put the staging arrays there), the streaming decode kernel spills nothing, the 16-row
    prefill kernel spills only in its predicated d = 240 form."""
    ks = kernel_notes(GEN, str(tmp_path))
    t64 = {n: v for n, v in ks.items() if "attn_prefill_tile64_kernel" in n}
    assert len(t64) >= 2 * (8 * 2 + 4 * 2)
    for n, v in t64.items():
        assert v["vgpr"] <= 256, (n, v)
        if "ELi7ELi64E" in n or "ELi8ELi64E" in n:   # d = 224 with 64-key tiles: ~19 registers spilled (40 bytes of scratch) and still 1.6 x the 32-key form (profiles/r05_generic_prefill_cfg.json); d = 256 takes 32-key tiles (its 64-key form, one spill, is an A/B option)
            assert v["scratch"] <= 64, (n, v)
        else:
            assert v["scratch"] == 0 and v["vspill"] == 0, (n, v)
    ad2 = {n: v for n, v in ks.items() if "attn_decode_anyd2_kernel" in n}
    assert len(ad2) >= 2 * 4 * 3 * 4
    for n, v in ad2.items():         # (the 4-wavefront, 4-q-head form at head size 128 parks 24 registers in the accumulator file: no memory traffic)
        assert v["scratch"] == 0, (n, v)
    for n, v in ks.items():
        if "attn_prefill_tile16_kernel" in n and "ELb1E" in n:      # the exact head-size instantiations
            assert v["scratch"] == 0 and v["vspill"] == 0, (n, v)
