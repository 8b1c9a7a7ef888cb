"""bindings/tp.py preflight -- the first-contact check bench.py --gpus N runs before its timed region -- on CPU: two gloo
processes drive it with a stand-in for the device library (the sums are done by torch.distributed itself), because what must hold on the first
8-GPU box cannot be run on a 1-GPU box: every rank takes the agreement of every (engine, size) pair exactly once whatever happened locally -- a wrong
sum, an error or a HANG on one rank fails that pair on every rank, later pairs are skipped alike, and the ranks' collectives stay in step."""
import multiprocessing as mp
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scenario, q_out):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
        import torch
        import torch.distributed as dist
        import tp
        dist.init_process_group("gloo", rank=rank, world_size=world)
        side = dist.new_group(backend="gloo")           # the stand-in's own "links": separate from the rendezvous preflight agrees over

        class Buf:
            def __init__(self, a):
                self.a, self.ptr = a, id(a)
                Fake.bufs[self.ptr] = self

            @classmethod
            def from_numpy(cls, a):
                return cls(np.array(a, copy=True))

            @classmethod
            def zeros(cls, shape, dtype):
                return cls(np.zeros(shape, dtype))

            def numpy(self, dtype, shape):
                return self.a.view(dtype).reshape(shape)

            def free(self):
                Fake.bufs.pop(self.ptr, None)

        class Lib:
            calls = 0

            def atoma_comm_set_mode(self, comm, mode):
                Fake.mode = mode
                return 0

            def atoma_allreduce_sum(self, comm, i, o, n, dtype, stream):
                Lib.calls += 1
                src, dst = Fake.bufs[i].a, Fake.bufs[o].a
                if scenario == "hang" and Fake.mode == 1 and rank == 1 and n * 2 == 1 << 20:
                    time.sleep(30)                       # the direct engine never returns on rank 1 at 1 MiB
                f = (src.astype(np.uint32) << 16).view(np.float32).copy()
                if scenario == "hang" and Fake.mode == 1 and n * 2 == 1 << 20:
                    return 0                             # (rank 0's peers are gone: it returns with its own data)
                t = torch.from_numpy(f)
                dist.all_reduce(t, group=side)
                if scenario == "wrong" and Fake.mode == 0 and rank == 0 and n == 8:
                    t[3] += 1.0                          # rank 0 reads a stale value at 16 B through RCCL
                dst[:] = (t.numpy().view(np.uint32) >> 16).astype(np.uint16)
                return 0

            def atoma_xgmi_capacity(self, x):
                return 1 << 40

        class Fake:
            bufs, mode = {}, 0
            DeviceBuffer, lib = Buf, Lib()

            @staticmethod
            def synchronize():
                pass

            @staticmethod
            def last_error():
                return "stand-in"
        logs = []
        res = tp.preflight(Fake, dist, rank, world, comm=object(), sizes=(16, 1 << 20, 1 << 21), log=logs.append, timeout_s=3.0)
        # the ranks must still be in step: one more collective over the rendezvous
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        q_out.put((rank, res, float(t.item()), logs))
    except Exception as e:
        import traceback
        q_out.put((rank, "error: " + traceback.format_exc() + repr(e), -1.0, []))
    finally:
        q_out.close()
        q_out.join_thread()                              # the result has left this process ...
        os._exit(0)                                      # ... which may still hold a worker thread asleep in the "hang"


def _run(scenario):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scenario, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(30)
    for r in res:
        assert not isinstance(r[1], str), r[1]
        assert r[2] == 1.0, "the ranks' collectives are out of step after the preflight"
    strip = lambda r: {e: {k: (v["ok"], v["us"]) for k, v in d.items()} for e, d in r.items()}     # (the "note" is the rank's own view: only the failing rank says why)
    assert strip(res[0][1]) == strip(res[1][1]), "every rank must report the same verdicts"
    return res[0][1], res[0][3]


def test_preflight_passes_and_times_every_pair():
    r, logs = _run("ok")
    for eng in ("rccl", "direct"):
        for size in ("16B", "1MiB", "2MiB"):
            assert r[eng][size]["ok"] and r[eng][size]["us"] > 0, r
    assert len(logs) == 6 and all("PASS" in l for l in logs)


def test_preflight_a_wrong_sum_on_one_rank_fails_the_pair_everywhere():
    r, logs = _run("wrong")
    assert r["rccl"]["16B"]["ok"] is False and r["rccl"]["1MiB"]["ok"] and r["direct"]["16B"]["ok"], r
    assert any("FAIL" in l for l in logs)


def test_preflight_a_hang_on_one_rank_fails_the_pair_and_skips_the_rest_on_every_rank():
    r, _ = _run("hang")
    assert r["rccl"]["2MiB"]["ok"] and r["direct"]["16B"]["ok"], r
    assert r["direct"]["1MiB"]["ok"] is False and r["direct"]["2MiB"]["ok"] is False, r
    assert "not run" in r["direct"]["2MiB"].get("note", "") and "hung" in r["direct"]["1MiB"].get("note", ""), r     # (rank 0's view: its own sum was wrong AND another rank hung)
