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


# ---- the in-launch merges' memory order, checked in the emitted ISA -------------------------------------------------------
# The arrival tickets are relaxed atomics (csrc/sync_ticket.h); what orders a partial before its ticket and a read after it is what the
# INSTRUCTIONS do: partials leave as sc1 (write-through, agent-scope) stores, the wavefront drains them (s_waitcnt vmcnt(0)) before the ticket
# atomic, and the last arriver reads them with sc1 loads (past its L1 / the XCD's L2).  The HIP / LLVM memory model does not promise that for
# relaxed atomics, so a compiler upgrade could drop or move an sc1 without any test noticing until a rare stale read -- these checks notice at
# build time.
def disassemble(obj, tmp):
    fb, co = os.path.join(tmp, "fb"), os.path.join(tmp, "co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co])
    text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
        elif cur is not None and "\t" in line:
            cur.append(line.split("//")[0].strip())
    return funcs


def _ticket_discipline(name, ins):
    """every ticket atomic (global_atomic_add_x2 behind a global_atomic_umax_x2: sync_arrive) has an `s_waitcnt vmcnt(0)` between the last
    sc1 store before it and itself"""
    adds = [i for i, t in enumerate(ins) if t.startswith("global_atomic_add_x2")]
    assert adds, name
    for a in adds:
        stores = [i for i in range(a) if ("store" in ins[i] and "sc1" in ins[i])]
        assert stores, f"{name}: no write-through store ahead of the ticket"
        between = ins[stores[-1]:a]
        assert any(re.match(r"s_waitcnt.*vmcnt\(0\)", t) for t in between), f"{name}: the partial stores are not drained before the ticket"


@pytest.mark.parametrize("obj,kernel,frag_of", [("linear_tile", "linear_tile_kernel", lambda nw, br: nw // 32), ("linear_wide", "linear_wide_kernel", lambda nw, br: (nw // 32) * (br // 64))])
def test_tile_kernels_publish_and_read_their_slabs_write_through(tmp_path, obj, kernel, frag_of):
    path = os.path.join(ROOT, "atoma-infer_amd", "build", obj + ".o")
    if not (os.path.exists(path) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("needs the built object and the ROCm llvm tools")
    ks = {n: v for n, v in disassemble(path, str(tmp_path)).items() if kernel in n}
    assert len(ks) >= 18
    for n, ins in ks.items():
        dims = [int(x) for x in re.findall(r"ELi(\d+)", n)]
        frags = frag_of(dims[0], dims[1] if len(dims) > 2 else 64)
        st = [t for t in ins if t.startswith("buffer_store_dwordx4")]
        ld = [t for t in ins if t.startswith("buffer_load_dwordx4")]
        # the ONLY 16-byte buffer accesses of these kernels are the fp32 slabs of the K-split merge: all of them write-through / read-through
        assert len(st) == frags and all("sc1" in t for t in st), (n, st[:2])
        assert len(ld) >= frags and all("sc1" in t for t in ld), (n, ld[:2])
        _ticket_discipline(n, ins)


def test_decode_merges_publish_and_read_their_pieces_write_through(tmp_path):
    if not (os.path.exists(OBJ) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("needs the built object and the ROCm llvm tools")
    fs = disassemble(OBJ, str(tmp_path))
    merges = {n: v for n, v in fs.items() if "decode_line_merge" in n}
    assert len(merges) >= 4
    for n, ins in merges.items():                       # the last arriver's reads of LSEs and rows: agent-scope loads
        sc1_loads = [t for t in ins if re.match(r"(flat|global)_load_dword", t) and "sc1" in t]
        assert len(sc1_loads) >= 3, (n, len(sc1_loads))
    # (decode_combine_kernel is a separate launch: the kernel boundary orders its plain loads behind the split launch's stores)
    ticketed = {n: v for n, v in fs.items() if "paged_decode" in n and "kernel" in n and any(t.startswith("global_atomic_add_x2") for t in v)}
    assert len(ticketed) >= 16
    for n, ins in ticketed.items():
        assert sum(1 for t in ins if "store" in t and "sc1" in t) >= 2, n
        _ticket_discipline(n, ins)
