"""The N > 1 path on CPU: two processes over gloo check that the head sharding bench.py uses, plus one
sum all-reduce of the row-parallel projection, reproduces the unsharded layer (llama_nccl.rs:139,153-171;
multi_gpu.rs:48-50), and that the unique-id bootstrap hands every rank the same bytes.  The GPU-side
collective itself is RCCL's (tests/test_host_ops_gpu.py exercises the C ABI with a world of one)."""
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


def _worker(rank, world, port, q_out):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
        import torch
        import torch.distributed as dist
        from oracle import attn_oracle as A
        from oracle.halfs import BF16, to_f32
        from util import rand_half, make_paged_cache
        import tp
        dist.init_process_group("gloo", rank=rank, world_size=world)
        uid = tp.broadcast_unique_id(dist, lambda: bytes(range(128)), rank)
        assert uid == bytes(range(128))
        # identical inputs on every rank (the reference clones one ExecuteModelRequest to every GPU thread)
        rng = np.random.default_rng(0)
        B, h, hk, d, page = 3, 8, 4, 64, 16
        lens = np.array([40, 17, 100], np.int32)
        kc, vc, bt = make_paged_cache(rng, 12, page, hk, d, BF16, lens)
        q = rand_half(rng, (B, 1, h, d), BF16)
        w_o = rng.standard_normal((h * d, 32)).astype(np.float32)       # o_proj, row-parallel over heads
        full = to_f32(A.flash_attn_kv_cache(q, kc, vc, d ** -0.5, BF16, bt, lens), BF16).reshape(B, h * d) @ w_o
        qs, ks = tp.head_shard(h, hk, rank, world)
        part = A.flash_attn_kv_cache(np.ascontiguousarray(q[:, :, qs]), np.ascontiguousarray(kc[:, :, ks]),
                                     np.ascontiguousarray(vc[:, :, ks]), d ** -0.5, BF16, bt, lens)
        y = to_f32(part, BF16).reshape(B, -1) @ w_o[qs.start * d: qs.stop * d]
        t = torch.from_numpy(y.astype(np.float32))
        dist.all_reduce(t)                                                  # the path's one exchange step
        err = float(np.abs(t.numpy() - full).max())
        # max-over-ranks timing reduction as bench.py does it
        tm = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        q_out.put((rank, err, float(tm.item())))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        q_out.put((rank, repr(e), -1.0))


def test_two_rank_head_sharding_reproduces_unsharded_layer():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, err, tmax in res:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert err < 1e-3, err
        assert tmax == 2.0


def test_head_shard_partitions_all_heads():
    sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
    import tp
    for h, hk, world in ((32, 8, 1), (32, 8, 2), (32, 8, 8), (64, 8, 4)):
        qh, kh = [], []
        for r in range(world):
            qs, ks = tp.head_shard(h, hk, r, world)
            qh += list(range(h))[qs]
            kh += list(range(hk))[ks]
            assert (qs.stop - qs.start) // (ks.stop - ks.start) == h // hk      # GQA group stays whole
        assert qh == list(range(h)) and kh == list(range(hk))
    import pytest
    with pytest.raises(ValueError):
        tp.head_shard(32, 8, 0, 3)
