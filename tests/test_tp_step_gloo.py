"""The tensor-parallel decode STEP on CPU: two processes over gloo, each running tests/tp_oracle_step.py on its shard
(tp.shard_config / tp.shard_weights -- the same helpers the device drivers use) with a sum all-reduce after the o and the
down projection, against the unsharded step.  Checks the host-side sharding logic of the N > 1 path: which rows / columns
of which weight a rank owns, which kv heads its cache holds, where the two exchanges sit (llama_nccl.rs:139,153-171,195;
multi_gpu.rs:20-57).  No GPU."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
        if p not in sys.path:
            sys.path.insert(0, p)


class Cfg:
    """The fields of tools/decode_step.Config (that module loads the HIP library; this test must run without it)."""

    def __init__(self, layers, hidden, heads, kv_heads, head_dim, intermediate, vocab, page=16, eps=1e-5, theta=500000.0, max_pos=8192):
        self.layers, self.hidden, self.h, self.hk, self.d = layers, hidden, heads, kv_heads, head_dim
        self.inter, self.vocab, self.page, self.eps, self.theta, self.max_pos = intermediate, vocab, page, eps, theta, max_pos
        self.qkv = (heads + 2 * kv_heads) * head_dim


def _problem():
    _setup_paths()
    from oracle import cache_oracle as CO
    from oracle import norm_rope_oracle as NR
    from oracle.halfs import BF16, from_f32
    from util import rand_half
    rng = np.random.default_rng(5)
    cfg = Cfg(2, 256, 8, 4, 32, 512, 301, max_pos=128)
    B = 4
    ctx = np.array([0, 5, 16, 33])
    lens = (ctx + 1).astype(np.int32)
    blocks = [(int(L) + cfg.page - 1) // cfg.page for L in lens]
    nb = sum(blocks) + 2
    perm = rng.permutation(nb)
    bt = np.zeros((B, max(blocks)), np.int32)
    p0 = 0
    for i, n in enumerate(blocks):
        bt[i, :n] = perm[p0:p0 + n]
        p0 += n
    H, I = cfg.hidden, cfg.inter
    r = lambda shape, scale: from_f32((rng.standard_normal(shape) * scale).astype(np.float32), BF16)
    near1 = lambda: from_f32((1 + 0.1 * rng.standard_normal(H)).astype(np.float32), BF16)
    host = dict(emb=r((cfg.vocab, H), 1.0), norm1=[near1() for _ in range(cfg.layers)], wqkv=[r((cfg.qkv, H), H ** -0.5) for _ in range(cfg.layers)],
                wo=[r((H, cfg.h * cfg.d), (cfg.h * cfg.d) ** -0.5) for _ in range(cfg.layers)], norm2=[near1() for _ in range(cfg.layers)],
                wgu=[r((2 * I, H), H ** -0.5) for _ in range(cfg.layers)], wdown=[r((H, I), I ** -0.5) for _ in range(cfg.layers)],
                norm_f=near1(), lm_head=r((cfg.vocab, H), H ** -0.5))
    kc = [rand_half(rng, (nb, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    vc = [rand_half(rng, (nb, cfg.page, cfg.hk, cfg.d), BF16) for _ in range(cfg.layers)]
    ids = rng.integers(0, cfg.vocab, B)
    slots = np.array([CO.slot_mapping_for(bt[i], int(ctx[i]), int(ctx[i]) + 1, cfg.page)[0] for i in range(B)], np.int64)
    cos, sin = NR.rope_table(cfg.max_pos, cfg.d, cfg.theta, BF16)
    return cfg, host, kc, vc, ids, ctx, slots, lens, bt, cos, sin


def _worker(rank, world, port, q_out):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        _setup_paths()
        import torch
        import torch.distributed as dist
        import tp
        import tp_oracle_step as TS
        from oracle.halfs import BF16, to_f32, from_f32
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cfg, host, kc, vc, ids, ctx, slots, lens, bt, cos, sin = _problem()          # identical on every rank (seeded)
        scfg = tp.shard_config(cfg, world)
        w = tp.shard_weights(host, cfg, rank, world)
        _, ks = tp.head_shard(cfg.h, cfg.hk, rank, world)
        kcs = [np.ascontiguousarray(a[:, :, ks]) for a in kc]
        vcs = [np.ascontiguousarray(a[:, :, ks]) for a in vc]

        def allreduce(bits):          # sum of the ranks' bf16 tensors in fp32, one rounding (multi_gpu.rs:141-179)
            t = torch.from_numpy(to_f32(bits, BF16).copy())
            dist.all_reduce(t)
            return from_f32(t.numpy(), BF16)
        logits, trace = TS.decode_step(scfg, w, kcs, vcs, ids, ctx, slots, lens, bt, cos, sin, allreduce)
        # every rank must hold the same logits: gather rank 1's on rank 0 through an all-reduce of (rank == 1) * logits
        mine = torch.from_numpy(to_f32(logits, BF16).copy())
        other = mine.clone() if rank == 1 else torch.zeros_like(mine)
        dist.all_reduce(other)
        same = bool(torch.equal(other, mine))
        q_out.put((rank, "ok", logits if rank == 0 else None, same, [t["o"] for t in trace] if rank == 0 else None, kcs[0] if rank == 0 else None))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        import traceback
        q_out.put((rank, "error: " + traceback.format_exc() + repr(e), None, False, None, None))


def test_two_rank_oracle_step_matches_unsharded():
    _setup_paths()
    import tp_oracle_step as TS
    import tp
    from oracle.halfs import BF16, to_f32
    world = 2
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = _free_port()
    procs = [ctx_mp.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=180)
        res[r[0]] = r
    for p in procs:
        p.join(30)
    for r in range(world):
        assert res[r][1] == "ok", res[r][1]
        assert res[r][3], "the ranks ended the step with different logits"
    cfg, host, kc, vc, ids, ctx, slots, lens, bt, cos, sin = _problem()
    kc_ref = [a.copy() for a in kc]
    logits_ref, trace_ref = TS.decode_step(cfg, host, kc_ref, vc, ids, ctx, slots, lens, bt, cos, sin)
    got, ref = to_f32(res[0][2], BF16), to_f32(logits_ref, BF16)
    # sharding moves rounding points (each rank rounds its partial projection before the sum): a few bf16 ulps on O(1) values
    assert np.abs(got - ref).max() < 0.06, np.abs(got - ref).max()
    assert np.array_equal(got.argmax(1), ref.argmax(1))
    o_tp, o_ref = to_f32(res[0][4][0], BF16), to_f32(trace_ref[0]["o"], BF16)
    assert (np.abs(o_tp - o_ref) <= 0.01 + 2.0 ** -6 * np.abs(o_ref)).all()      # first layer: identical inputs, only the split sum differs
    # rank 0's cache holds exactly kv heads [0, hk / 2) of the unsharded cache after the step's write
    _, ks = tp.head_shard(cfg.h, cfg.hk, 0, world)
    assert np.array_equal(res[0][5], kc_ref[0][:, :, ks])


def test_shard_helpers_cover_every_weight_exactly_once():
    _setup_paths()
    import tp
    cfg, host, *_ = _problem()
    world = 4
    shards = [tp.shard_weights(host, cfg, r, world) for r in range(world)]
    sc = tp.shard_config(cfg, world)
    assert (sc.h, sc.hk, sc.inter, sc.hidden, sc.vocab) == (cfg.h // 4, cfg.hk // 4, cfg.inter // 4, cfg.hidden, cfg.vocab)
    d = cfg.d
    for l in range(cfg.layers):
        full = host["wqkv"][l]
        q = np.concatenate([s["wqkv"][l][:sc.h * d] for s in shards])
        k = np.concatenate([s["wqkv"][l][sc.h * d:(sc.h + sc.hk) * d] for s in shards])
        v = np.concatenate([s["wqkv"][l][(sc.h + sc.hk) * d:] for s in shards])
        assert np.array_equal(np.concatenate([q, k, v]), full)
        assert np.array_equal(np.concatenate([s["wo"][l] for s in shards], 1), host["wo"][l])
        assert np.array_equal(np.concatenate([s["wdown"][l] for s in shards], 1), host["wdown"][l])
        gate = np.concatenate([s["wgu"][l][:sc.inter] for s in shards])
        up = np.concatenate([s["wgu"][l][sc.inter:] for s in shards])
        assert np.array_equal(np.concatenate([gate, up]), host["wgu"][l])
    import pytest
    with pytest.raises(ValueError):
        tp.shard_config(cfg, 3)
