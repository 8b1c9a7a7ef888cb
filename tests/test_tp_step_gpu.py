"""A tensor-parallel decode step on the device: two ranks, each a DecodeStep over its weight / head / cache shard with the
direct xGMI all-reduce after the o and the down projection (llama_nccl.rs:139,195; multi_gpu.rs:48-50,141-179), both
placed on device 0 (own streams, own staging regions -- everything but the physical link), against the UNSHARDED step
on the same device, which tests/test_decode_step_gpu.py pins op by op to the oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from oracle import allreduce_oracle as AO
from oracle import cache_oracle as CO
from oracle.halfs import BF16, to_f32
from util import rand_half

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("graph,own,B,fused", [(False, False, 6, False), (True, False, 6, False), (True, True, 6, False), (True, True, 24, False), (True, True, 24, True), (False, False, 6, True)],
                         ids=["eager", "hipgraph", "hipgraph-own-projections", "hipgraph-own-projections-24-rows", "hipgraph-own-24-rows-allreduce+add+norm-in-one-launch",
                              "eager-allreduce+add+norm-in-one-launch"])
def test_tp_decode_step_matches_unsharded(gpu, world, graph, own, B, fused, monkeypatch):
    """own: the rank's projections on the library's own kernels (gate/up with SiLU.up inside; at 24 rows the LDS-DMA tile kernel with its
    in-launch K-split merge and the q/k/v projection with RoPE + cache write as its epilogue; tools/tp_step.py and bench.py run it so).
    world 8: the partition of configs[3] -- every rank one kv head and its q-head group, an eighth of the MLP, 7 peers in every all-reduce."""
    monkeypatch.setenv("ATOMA_XGMI_TIMEOUT_MS", "8000")
    import decode_step as DS
    import tp
    from test_allreduce_xgmi_gpu import make_ranks
    rng = np.random.default_rng(21)
    cfg = (DS.Config(layers=3, hidden=512, heads=8, kv_heads=2, head_dim=128, intermediate=1024, vocab=1008, page=16, max_pos=256) if world == 2 else
           DS.Config(layers=3, hidden=512, heads=16, kv_heads=8, head_dim=128, intermediate=2048, vocab=1008, page=16, max_pos=256))
    ctx = np.array([0, 17, 40, 64, 100, 130]) if B == 6 else np.sort(rng.integers(0, 131, B))
    lens = (ctx + 1).astype(np.int32)
    blocks = [(int(L) + cfg.page - 1) // cfg.page for L in lens]
    num_pages = sum(blocks) + 3
    perm = rng.permutation(num_pages)
    bt = np.zeros((B, max(blocks)), np.int32)
    p0 = 0
    for i, n in enumerate(blocks):
        bt[i, :n] = perm[p0:p0 + n]
        p0 += n
    host = DS.random_host_weights(rng, cfg)
    kc0 = [rand_half(rng, (num_pages, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    vc0 = [rand_half(rng, (num_pages, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    ids = rng.integers(0, cfg.vocab, B)
    slots = np.array([CO.slot_mapping_for(bt[i], int(ctx[i]), int(ctx[i]) + 1, cfg.page)[0] for i in range(B)], np.int64)

    # ---- the unsharded step ----
    st = gpu.Stream()
    full = DS.DecodeStep(cfg, B, num_pages, bt.shape[1], DS.upload_weights(cfg, host), st, keep_intermediates=True)
    for l in range(cfg.layers):
        full.kc[l].upload(kc0[l])
        full.vc[l].upload(vc0[l])
    full.set_inputs(ids, ctx, slots, lens, bt)
    full.run()
    st.synchronize()
    H = cfg.hidden
    logits_full = to_f32(full.logits.numpy(np.uint16, (B, cfg.vocab)), BF16)
    o_full = [t["o"].numpy(np.uint16, (B, H)) for (n, l, t) in full.trace[1:-1]]
    dn_full = [t["dn"].numpy(np.uint16, (B, H)) for (n, l, t) in full.trace[1:-1]]

    # ---- the ranks ----
    xs = make_ranks(gpu, world, 1 << 20)
    streams = [gpu.Stream() for _ in range(world)]
    scfg = tp.shard_config(cfg, world)
    steps = []
    live = [False]
    fused_calls = [0]
    try:
        for r in range(world):
            def allreduce(ptr, count, r=r):
                if live[0]:
                    assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], ptr, ptr, count, BF16, streams[r].s) == 0, gpu.last_error()
            def allreduce_norm(inp, res, wn, xo, no, rows, r=r):   # fused: the residual add and the RMSNorm ride in the all-reduce's launch
                if not (live[0] and fused):
                    return False
                assert gpu.lib.atoma_xgmi_allreduce_add_rms_norm(xs[r], inp, res, wn, xo, no, rows, cfg.hidden, cfg.hidden, cfg.hidden, cfg.hidden, cfg.eps, BF16, 0,
                                                                 streams[r].s) == 0, gpu.last_error()
                fused_calls[0] += 1
                return True
            w = DS.upload_weights(scfg, tp.shard_weights(host, cfg, r, world))
            s = DS.DecodeStep(scfg, B, num_pages, bt.shape[1], w, streams[r], keep_intermediates=not graph and not fused, allreduce=allreduce, fused_epilogues=own,
                              allreduce_norm=allreduce_norm)
            assert s.tp_own == own
            _, ks = tp.head_shard(cfg.h, cfg.hk, r, world)
            for l in range(cfg.layers):                        # the rank's KV cache holds its kv heads (worker.rs:584-591)
                s.kc[l].upload(np.ascontiguousarray(kc0[l][:, :, ks]))
                s.vc[l].upload(np.ascontiguousarray(vc0[l][:, :, ks]))
            s.set_inputs(ids, ctx, slots, lens, bt)            # identical metadata on every rank (model_executor.rs:531-542)
            steps.append(s)
        # Both ranks are driven from this one host thread.  The vendor GEMM behind the 6-row projections synchronises with
        # the stream the first time it sees a shape; that must not happen while rank 0's all-reduce waits on the device for
        # kernels of rank 1 that this thread has not enqueued yet: a first pass with the exchange switched off.
        for s in steps:
            s.run()
        for r in range(world):
            streams[r].synchronize()
        live[0] = True
        for r in range(world):                                 # the warm pass wrote the step's K/V rows already (same values); intermediates are rebuilt
            steps[r].trace = []
        if graph:
            # scratch of each stream sized explicitly (atoma_warmup); one eager step so that the vendor GEMM behind the
            # 6-row projections has its plan and workspace (it cannot create them during capture) -- it rewrites the same
            # cache slots with the same values
            graphs = []
            for r in range(world):
                assert gpu.lib.atoma_warmup(streams[r].s, B, scfg.h, scfg.hk, scfg.d, bt.shape[1] * cfg.page, 64 << 20) == 0, gpu.last_error()
            for s in steps:
                s.run()
            for r in range(world):
                streams[r].synchronize()
            for r in range(world):
                with gpu.Graph.capture(streams[r]) as g:
                    steps[r].run()
                graphs.append(g)
            for g in graphs:
                g.launch()
        else:
            for s in steps:                                    # rank 0's kernels wait on the device for rank 1's
                s.run()
        for r in range(world):
            streams[r].synchronize()
            assert gpu.lib.atoma_xgmi_status(xs[r]) == 0, f"rank {r}: an all-reduce wait timed out"
        logits = [s.logits.numpy(np.uint16, (B, cfg.vocab)) for s in steps]
        for r in range(1, world):
            assert np.array_equal(logits[0], logits[r]), f"ranks 0 and {r} disagree: the all-reduce must leave bit-identical activations everywhere"
        l32 = to_f32(logits[0], BF16)
        # sharding moves rounding points (partial projections are rounded before they are summed): a few bf16 ulps on O(1) logits
        assert np.abs(l32 - logits_full).max() < 0.08, np.abs(l32 - logits_full).max()
        ids_tp, ids_full = steps[0].next_ids.numpy(np.int32, (B,)), full.next_ids.numpy(np.int32, (B,))
        top2 = np.sort(logits_full, 1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 0.1                  # rows whose winner is not a near tie
        assert clear.any() and np.array_equal(ids_tp[clear], ids_full[clear])
        assert (fused_calls[0] > 0) == fused
        if fused:      # the same step with the two separate calls: the fused launch must not move a bit
            fused_logits = logits[0].copy()
            live[0] = False
            for s_ in steps:
                s_.allreduce_norm = None
            live[0] = True
            for s_ in steps:
                s_.run()
            for r in range(world):
                streams[r].synchronize()
            assert np.array_equal(steps[0].logits.numpy(np.uint16, (B, cfg.vocab)), fused_logits), "all-reduce + add + norm in one launch differs from the three calls"
        if not graph and not fused:
            for li, (n, l, t) in enumerate(steps[0].trace[1:-1]):
                # the summed projection outputs against the unsharded projection (same inputs up to earlier rounding)
                for key, ref in (("o", o_full[li]), ("dn", dn_full[li])):
                    got = to_f32(t[key].numpy(np.uint16, (B, H)), BF16)
                    r32 = to_f32(ref, BF16)
                    # (three to four bf16 ulps at |x| ~ 1 after three layers: 0.0273 observed with the matrix-core decode kernel, 0.02 with dot2)
                    excess = (np.abs(got - r32) - (0.03 + 2.0 ** -5 * np.abs(r32))).max()
                    assert excess <= 0, f"layer {l} {key}: {excess:.4f} beyond 0.03 + 2^-5 |ref| (max |diff| {np.abs(got - r32).max():.4f})"
    finally:
        for x in xs:
            gpu.lib.atoma_xgmi_destroy(x)


@pytest.mark.parametrize("world", [2, 8])
def test_tp_prefill_chunk_matches_unsharded(gpu, world, monkeypatch):
    """The prefill phase of the tensor-parallel step (llama_nccl.rs:118-200 on prompt tokens): every rank projects all T tokens onto its
    q / kv heads, runs causal prefill attention over ITS heads, and the [T, hidden] outputs of the o and the down projection are
    all-reduced -- two prompts of 96 tokens here, against the unsharded PrefillStep on the same weights.  (The full shapes -- T = 4096,
    hidden 8192, a 64 MiB message among 8 ranks -- run in tests/test_tp_world8_gpu.py.)"""
    monkeypatch.setenv("ATOMA_XGMI_TIMEOUT_MS", "8000")
    import decode_step as DS
    import tp
    from test_allreduce_xgmi_gpu import make_ranks
    rng = np.random.default_rng(33)
    cfg = DS.Config(layers=2, hidden=512, heads=16, kv_heads=8, head_dim=128, intermediate=2048, vocab=1008, page=16, max_pos=256)
    T, n = 192, 2
    pages = T // cfg.page
    host = DS.random_host_weights(rng, cfg)
    ids = rng.integers(0, cfg.vocab, T)
    slots = (rng.permutation(pages)[np.arange(T) // cfg.page] * cfg.page + np.arange(T) % cfg.page).astype(np.int64)

    st = gpu.Stream()
    fd = DS.DecodeStep(cfg, 1, pages + 1, pages, DS.upload_weights(cfg, host), st)
    full = DS.PrefillStep(cfg, T, fd, st, prompts=n)
    full.set_inputs(ids, slots)
    full.run()
    st.synchronize()
    logits_full = to_f32(full.logits.numpy(np.uint16, (n, cfg.vocab)), BF16)

    xs = make_ranks(gpu, world, 1 << 20)
    streams = [gpu.Stream() for _ in range(world)]
    scfg = tp.shard_config(cfg, world)
    live = [False]
    steps = []
    try:
        for r in range(world):
            def allreduce(ptr, count, r=r):
                if live[0]:
                    assert count == T * cfg.hidden
                    assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], ptr, ptr, count, BF16, streams[r].s) == 0, gpu.last_error()
            d = DS.DecodeStep(scfg, 1, pages + 1, pages, DS.upload_weights(scfg, tp.shard_weights(host, cfg, r, world)), streams[r])
            p = DS.PrefillStep(scfg, T, d, streams[r], prompts=n, allreduce=allreduce)
            p.set_inputs(ids, slots)
            steps.append(p)
        for p in steps:                                        # vendor-GEMM plans with the exchange off (see the decode test)
            p.run()
        for r in range(world):
            streams[r].synchronize()
        live[0] = True
        for p in steps:
            p.run()
        for r in range(world):
            streams[r].synchronize()
            assert gpu.lib.atoma_xgmi_status(xs[r]) == 0, f"rank {r}: an all-reduce wait timed out"
        logits = [p.logits.numpy(np.uint16, (n, cfg.vocab)) for p in steps]
        for r in range(1, world):
            assert np.array_equal(logits[0], logits[r]), f"ranks 0 and {r} disagree"
        assert np.abs(to_f32(logits[0], BF16) - logits_full).max() < 0.08
        # the rank's KV cache holds the prompt's K rows of ITS kv heads: the unsharded cache's columns (same projection rows of the same
        # input, same RoPE; the vendor GEMM accumulates a [T, 1280]-wide product in another order than a [T, 160]-wide one: an ulp or two)
        _, ks = tp.head_shard(cfg.h, cfg.hk, world - 1, world)
        kfull = to_f32(fd.kc[0].numpy(np.uint16, (pages + 1, cfg.page, cfg.hk, cfg.d))[:, :, ks], BF16)
        kshard = to_f32(steps[-1].kc[0].numpy(np.uint16, (pages + 1, cfg.page, scfg.hk, cfg.d)), BF16)
        assert np.abs(kfull).max() > 0.5 and (np.abs(kfull - kshard) <= 1e-2 + 2.0 ** -6 * np.abs(kfull)).all()
    finally:
        for x in xs:
            gpu.lib.atoma_xgmi_destroy(x)
