"""The hand-scheduled prefill kernel (csrc/prefill_asm.hip) without a GPU: the generated instruction list (tools/pfasm/kernel.py)
is executed by the functional simulator (tools/pfasm/sim.py -- adversarial load timing, MFMA hazard distances) one workgroup at
a time and compared with the oracle's own-schedule restatement (oracle/attn_oracle.py attend_prefill_online; `prescale` = the fast
variant's one rounding of scale.log2(e).Q) at 1e-3 + 1 ulp, and with the f32 definition at the tolerance of tests/util.py.  What
the simulator accepts is, instruction for instruction, what hipcc assembles into libatoma_hip.so."""
import numpy as np
import pytest

from oracle import attn_oracle as A
from oracle.halfs import BF16, F16, to_f32, from_f32
from tools.pfasm import harness as H
from tools.pfasm import kernel as K
from util import rand_half, assert_close, make_paged_cache, ATOL_VS_F32

D = 128


def run_case(lens, h, hk, causal, dtype=BF16, lens_k=None, exact=False, seed=1, spike=None, auto=False, **kw):
    """exact = True / False: every block on that arithmetic; auto: the kernel's rule (first row of the block sees < 512 keys -> exact)"""
    rng = np.random.default_rng(seed)
    lens = np.array(lens, np.int32)
    lk = lens if lens_k is None else np.array(lens_k, np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cuk = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
    q = rand_half(rng, (int(cu[-1]), h, D), dtype)
    k, v = rand_half(rng, (int(cuk[-1]), hk, D), dtype), rand_half(rng, (int(cuk[-1]), hk, D), dtype)
    if spike is not None:                      # key `spike[0]` = the query of row `spike[1]`: a score ~ d far above the others
        k[spike[0]] = q[spike[1], ::(h // hk)][:hk]
    out, lse, stats = H.prefill_varlen(q, k, v, cu, cuk, D ** -0.5, causal, "bf16" if dtype == BF16 else "f16", want_lse=True,
                                       exact=None if auto else exact, **kw)
    qf, kf, vf = to_f32(q, dtype), to_f32(k, dtype), to_f32(v, dtype)
    for b in range(len(lens)):
        s0, s1, k0, k1 = int(cu[b]), int(cu[b + 1]), int(cuk[b]), int(cuk[b + 1])
        if s1 == s0:
            continue
        if not auto:
            own = from_f32(A.attend_prefill_online(qf[s0:s1], kf[k0:k1], vf[k0:k1], np.float32(D ** -0.5), causal, dtype, prescale=not exact), dtype)
            assert_close(out[s0:s1], own, dtype, atol=1e-3, what=f"sequence {b} vs the kernel's own schedule")
        if (exact or auto) and k1 > k0:
            ref, want_lse = A.attend_rows(qf[s0:s1], kf[k0:k1], vf[k0:k1], np.float32(D ** -0.5), causal=causal)
            assert_close(out[s0:s1], from_f32(ref, dtype), dtype, atol=ATOL_VS_F32[dtype], what=f"sequence {b} vs the f32 definition")
            fin = np.isfinite(want_lse)
            # the fast arithmetic's one rounding of scale.log2(e).Q moves the log-sum-exp by ~1e-3 (no caller reads it: SURVEY Q4)
            assert np.allclose(lse[:, s0:s1][fin], want_lse[fin], rtol=1e-4, atol=2.5e-3 if auto else 1e-4)
            assert np.all(np.isposinf(lse[:, s0:s1][~fin]))
    return stats


@pytest.mark.parametrize("exact", [False, True], ids=["fast", "exact"])
@pytest.mark.parametrize("causal", [True, False])
def test_ragged_lengths_around_every_boundary(exact, causal):
    """1, 31..33, 63..65 tokens, a 256-row block with a ragged tail, GQA group 2: masked / single-slot / idle variants all run"""
    run_case([1, 31, 33, 64, 65, 290], 2, 1, causal, exact=exact)


@pytest.mark.parametrize("late,reverse", [(True, False), (False, True)], ids=["dma-lands-late", "dma-lands-early-waves-reversed"])
def test_three_blocks_both_dma_timings(late, reverse):
    """700 rows = three 256-row blocks (1, 8 and 11 K/V tiles): the steady-state loop, ring wrap-around, every barrier; the
    LDS-DMA lands at the latest / earliest legal time"""
    run_case([700], 1, 1, True, late_dma=late, reverse=reverse)


def test_two_persistent_workgroups_and_the_kernels_own_choice_of_arithmetic():
    """two workgroups share the plan table round-robin; per block the entry's flag picks the arithmetic (first row sees < 512 keys ->
    exact, else fast): 700 rows = two exact blocks and a fast one walked by the same workgroups"""
    run_case([700, 300, 64], 1, 1, True, auto=True, g=2)
    run_case([600], 2, 1, False, auto=True, g=3)


def test_split_request_variant_of_the_block_boundary(monkeypatch):
    """PFA_SPLIT_REQ=1 (an experiment kept in the generator: K(0), K(1) ride in the epilogue, V(0), K(2) under the S(0) products, V(1) with
    K(3) around the mask / max sections -- measured level, profiles/r04_prefill_split_requests_ab.txt): another order of the same requests,
    other wait counts; both DMA timings, idle wavefronts, empty entries, a block without any key"""
    harness = H
    monkeypatch.setenv("PFA_SPLIT_REQ", "1")
    saved = dict(harness._progs)
    harness._progs.clear()
    try:
        run_case([700, 33, 300], 2, 1, True, auto=True, g=2)
        run_case([700], 1, 1, True, late_dma=False, reverse=True)
        run_case([100, 40, 70], 1, 1, True, lens_k=[400, 10, 0], exact=True)
    finally:
        harness._progs.clear()
        harness._progs.update(saved)


def test_f16_and_more_queries_than_keys_and_empty():
    run_case([513], 1, 1, True, dtype=F16)
    run_case([100, 40, 70], 1, 1, True, lens_k=[400, 10, 0], exact=True)
    run_case([100, 40], 1, 1, False, lens_k=[400, 10], exact=True)


def test_block_without_keys_between_two_blocks():
    """A block that sees no key at all FOLLOWED by a normal block of the same persistent workgroup (round 6, found by tests/fuzz_parity.py on the
    GPU: until then the tile-less block left the ring's read state half rewound and the next block read its K fragments from the wrong slot):
    the kv_cache shape that found it (3 query rows per sequence, a sequence without keys in the middle), empty sequences between prompts
    walked by two workgroups, and a causal sequence with more queries than keys (its first 256 rows see nothing) in front of another one."""
    run_case([3, 3, 3], 1, 1, False, lens_k=[4, 0, 30], exact=True)
    run_case([100, 40, 70, 300], 2, 1, True, lens_k=[400, 0, 0, 310], auto=True, g=2)
    run_case([64, 600, 64], 1, 1, True, lens_k=[64, 100, 64], auto=True)


def test_forced_late_raise_of_the_reference():
    """a key far above the others in the 4th tile of the last row (cdna guide T13 / rule 26): the rescale block with alpha != 1"""
    for exact in (False, True):
        base = run_case([300], 2, 1, True, exact=exact)
        st = run_case([300], 2, 1, True, exact=exact, spike=(200, 299))
        assert st["v_accvgpr_read_b32"] >= base["v_accvgpr_read_b32"] + 64     # O of a slot was read back for a rescale, not only by the epilogue


@pytest.mark.parametrize("page", [16, 64])
def test_paged_prefix(page):
    """prefix / chunked prefill over the paged cache: 90 new rows over 333 cached + new keys, random block table"""
    rng = np.random.default_rng(page)
    h, hk = 2, 1
    lens_q, lens_k = np.array([90, 17], np.int32), np.array([333, 81], np.int32)
    nb = int(sum((x + page - 1) // page for x in lens_k)) + 3
    kc, vc, bt = make_paged_cache(rng, nb, page, hk, D, BF16, lens_k)
    # unwritten slots of the last pages hold NaN patterns: they must not reach the output (P = 0 times NaN)
    for b, L in enumerate(lens_k):
        last = bt[b, (L - 1) // page]
        kc[last, (L - 1) % page + 1:] = 0x7FC0
        vc[last, (L - 1) % page + 1:] = 0x7FC0
    cu = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.int32)
    cuk = np.concatenate([[0], np.cumsum(lens_k)]).astype(np.int32)
    q = rand_half(rng, (int(cu[-1]), h, D), BF16)
    out, _, _ = H.prefill_varlen(q, kc, vc, cu, cuk, D ** -0.5, True, "bf16", block_table=bt, page_size=page, exact=True)
    qf = to_f32(q, BF16)
    for b in range(2):
        rows = [bt[b, i // page] * page + i % page for i in range(lens_k[b])]
        kf, vf = to_f32(kc.reshape(-1, hk, D)[rows], BF16), to_f32(vc.reshape(-1, hk, D)[rows], BF16)
        own = from_f32(A.attend_prefill_online(qf[cu[b]:cu[b + 1]], kf, vf, np.float32(D ** -0.5), True, BF16), BF16)
        assert_close(out[cu[b]:cu[b + 1]], own, BF16, atol=1e-3, what=f"paged, page {page}, sequence {b}")


def test_every_iteration_variant_is_reached():
    """the 15 iteration blocks + the idle one: each is entered by at least one of the shapes above (label coverage by simulation)"""
    prog, _ = K.build("bf16", False)
    labels = {i.ops[0]: n for n, i in enumerate(prog.ins) if i.op == "label"}
    its = sorted(l for l in labels if l.startswith("IT_"))
    assert len(its) == 17 and len([l for l in labels if l.startswith("ITX_")]) == 16
    from tools.pfasm.sim import Workgroup
    seen = set()
    orig = Workgroup.step

    def step(self, w):
        i = self.ins[w.pc] if w.pc < len(self.ins) else None
        if i is not None and i.op == "label" and i.ops[0].startswith("IT_"):
            seen.add(i.ops[0])
        return orig(self, w)
    Workgroup.step = step
    try:
        run_case([1, 33, 65, 290, 700], 1, 1, True)
        run_case([200, 200], 1, 1, False, lens_k=[130, 100])         # both slots end together, after an even / odd number of tiles
        run_case([160], 1, 1, True, lens_k=[96])                      # fewer keys than rows: slot 0 ends on an even tile, the next is masked
        run_case([256], 1, 1, True, lens_k=[288])                     # slot 0 ends on an odd tile, the next tile is plain
    finally:
        Workgroup.step = orig
    assert seen == set(its), sorted(set(its) - seen)


def test_schedule_density_of_the_plain_iteration():
    """the plain iteration (both slots, no mask): 64 MFMAs, at most 5 fillers in any MFMA gap, under 4.7 on average"""
    _, b = K.build("bf16", False)
    for exact, bound in ((False, 4.7), (True, 5.7)):
        l1, l2 = b.sched_log["ITX_0_220" if exact else "IT_0_220"]
        assert len(l1) == 33 and len(l2) == 33
        assert max(l1 + l2) <= (7 if exact else 6)
        assert sum(l1 + l2) / 64.0 <= bound
