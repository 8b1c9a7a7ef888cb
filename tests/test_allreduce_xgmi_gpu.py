"""The direct xGMI all-reduce (atoma_xgmi_*) on ONE device: several ranks, each with its own staging region and stream,
all placed on device 0.  That exercises everything but the physical link: region layout, handle exchange, peer mapping
(raw pointers inside a process, HIP IPC between processes), the flag protocol, both kernels, graph replay, timeouts.
Results are compared bit-for-bit with the rank-order fp32 sum (oracle/allreduce_oracle.py) and within one ulp with the
exactly rounded sum.  With >= 2 devices visible the same tests also run one rank per device, and against RCCL."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from oracle import allreduce_oracle as AO
from oracle.halfs import F16, BF16, to_f32
from util import rand_half

pytestmark = pytest.mark.gpu
F32 = 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_ranks(gpu, world, max_bytes, devices=None):
    devices = devices or [0] * world
    xs = []
    for r in range(world):
        h = C.c_void_p()
        assert gpu.lib.atoma_xgmi_create(C.byref(h), r, world, devices[r], max_bytes) == 0, gpu.last_error()
        xs.append(h)
    blobs = (C.c_uint8 * (128 * world))()
    for r in range(world):
        one = (C.c_uint8 * 128)()
        assert gpu.lib.atoma_xgmi_handle(xs[r], one) == 0, gpu.last_error()
        C.memmove(C.addressof(blobs) + 128 * r, one, 128)
    for r in range(world):
        gpu.set_device(devices[r])
        assert gpu.lib.atoma_xgmi_connect(xs[r], blobs) == 0, gpu.last_error()
    gpu.set_device(devices[0])
    return xs


def data(rng, world, count, dtype):
    if dtype == F32:
        return [rng.standard_normal(count).astype(np.float32) for _ in range(world)]
    return [rand_half(rng, (count,), dtype) for _ in range(world)]


def ulps(a, b):
    a, b = a.astype(np.int32), b.astype(np.int32)
    key = lambda x: np.where(x & 0x8000, 0x8000 - x, x)
    return np.abs(key(a) - key(b))


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("dtype", [BF16, F16, F32])
def test_virtual_ranks_one_shot_and_two_shot(gpu, world, dtype, monkeypatch):
    """world 8 = the communicator of configs[3] (TP = 8): 7 peers per rank, two-shot ownership S / 8 (8 streams of this process meet on
    device 0: tests/conftest.py gives every stream its own hardware queue)."""
    monkeypatch.setenv("ATOMA_XGMI_TIMEOUT_MS", "8000")
    monkeypatch.setenv("ATOMA_XGMI_ONESHOT_MAX", str(1 << 20))      # (the 1 MiB decode message of the 70B step through BOTH kernels)
    rng = np.random.default_rng(world * 10 + dtype)
    esz = 4 if dtype == F32 else 2
    xs = make_ranks(gpu, world, 4 << 20)
    streams = [gpu.Stream() for _ in range(world)]
    try:
        # 16 B, a ragged count (not a multiple of the world size in vectors), the 70B decode message (1 MiB), > one-shot
        # limit, and > capacity (cut into pieces); every size through both kernels where it fits
        for count_bytes, modes in ((16, (0, 1, 2)), (16 * 37, (1, 2)), (64 * 8192 * 2, (0, 1, 2)), (3 << 20, (0, 2)), (9 << 20, (0,))):
            count = count_bytes // esz
            parts = data(rng, world, count, dtype)
            want = AO.allreduce_sum(parts, dtype, "rank_order")
            exact = AO.allreduce_sum(parts, dtype, "exact")
            din = [gpu.DeviceBuffer.from_numpy(p) for p in parts]
            dout = [gpu.DeviceBuffer(count_bytes) for _ in range(world)]
            for mode in modes:
                for o in dout:
                    o.fill_bytes(0xFF)
                for r in range(world):       # all ranks are enqueued from this thread; their kernels meet on the device
                    rc = gpu.lib.atoma_xgmi_allreduce_sum_mode(xs[r], din[r].ptr, dout[r].ptr, count, dtype, mode, streams[r].s)
                    assert rc == 0, gpu.last_error()
                for r in range(world):
                    streams[r].synchronize()
                    assert gpu.lib.atoma_xgmi_status(xs[r]) == 0, f"rank {r}: a wait timed out"
                    got = dout[r].numpy(parts[0].dtype, (count,))
                    assert np.array_equal(got, want), f"bytes={count_bytes} mode={mode} rank {r}: differs from the rank-order fp32 sum"
                if dtype != F32:
                    assert ulps(want, exact).max() <= 1
        # in place, repeated back to back without host syncs (parity halves, sequence numbers)
        count = 64 * 8192
        parts = data(rng, world, count, dtype)
        bufs = [gpu.DeviceBuffer.from_numpy(p) for p in parts]
        cur = parts
        for it in range(5 if world < 8 else 3):
            for r in range(world):
                assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], bufs[r].ptr, bufs[r].ptr, count, dtype, streams[r].s) == 0, gpu.last_error()
            s = AO.allreduce_sum(cur, dtype, "rank_order")      # values grow by at most x world per round (3^5, 8^3): finite in f16 too
            cur = [s] * world
        for r in range(world):
            streams[r].synchronize()
            assert gpu.lib.atoma_xgmi_status(xs[r]) == 0
            assert np.array_equal(bufs[r].numpy(parts[0].dtype, (count,)), cur[0]), f"in-place chain, rank {r}"
    finally:
        for x in xs:
            gpu.lib.atoma_xgmi_destroy(x)


@pytest.mark.parametrize("world", [2, 8])
def test_virtual_ranks_prefill_message_64_mib(gpu, world, monkeypatch):
    """The all-reduce of configs[3]'s prefill chunk: [4096, 8192] bf16 = 64 MiB after the o and the down projection
    (llama_nccl.rs:139,195), in ONE launch per rank (capacity 64 MiB: rank r owns 8 MiB of it at world 8), by the size rule (two-shot),
    out of place and then in place on the result."""
    monkeypatch.setenv("ATOMA_XGMI_TIMEOUT_MS", "20000")
    count = 4096 * 8192
    rng = np.random.default_rng(64 + world)
    xs = make_ranks(gpu, world, count * 2)
    streams = [gpu.Stream() for _ in range(world)]
    try:
        assert gpu.lib.atoma_xgmi_capacity(xs[0]) == count * 2
        parts = data(rng, world, count, BF16)
        want = AO.allreduce_sum(parts, BF16, "rank_order")
        din = [gpu.DeviceBuffer.from_numpy(p) for p in parts]
        dout = [gpu.DeviceBuffer(count * 2) for _ in range(world)]
        for r in range(world):
            assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], din[r].ptr, dout[r].ptr, count, BF16, streams[r].s) == 0, gpu.last_error()
        for r in range(world):
            assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], dout[r].ptr, dout[r].ptr, count, BF16, streams[r].s) == 0, gpu.last_error()
        again = AO.allreduce_sum([want] * world, BF16, "rank_order")
        assert ulps(want, AO.allreduce_sum(parts, BF16, "exact")).max() <= 1
        for r in range(world):
            streams[r].synchronize()
            assert gpu.lib.atoma_xgmi_status(xs[r]) == 0, f"rank {r}: a wait timed out"
            assert np.array_equal(dout[r].numpy(np.uint16, (count,)), again), f"rank {r}: differs from the rank-order fp32 sum"
    finally:
        for x in xs:
            gpu.lib.atoma_xgmi_destroy(x)


@pytest.mark.parametrize("world", [2, 8])
def test_virtual_ranks_graph_replay(gpu, world, monkeypatch):
    """The call counter lives in device memory, so a captured all-reduce replays correctly any number of times."""
    monkeypatch.setenv("ATOMA_XGMI_TIMEOUT_MS", "8000")
    count = 256 * 4096
    rng = np.random.default_rng(3)
    xs = make_ranks(gpu, world, 4 << 20)
    streams = [gpu.Stream() for _ in range(world)]
    try:
        parts = data(rng, world, count, BF16)
        din = [gpu.DeviceBuffer.from_numpy(p) for p in parts]
        dout = [gpu.DeviceBuffer(count * 2) for _ in range(world)]
        graphs = []
        for r in range(world):
            with gpu.Graph.capture(streams[r]) as g:
                assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], din[r].ptr, dout[r].ptr, count, BF16, streams[r].s) == 0, gpu.last_error()
                assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], dout[r].ptr, dout[r].ptr, count, BF16, streams[r].s) == 0, gpu.last_error()
            graphs.append(g)
        once = AO.allreduce_sum(parts, BF16)
        want = AO.allreduce_sum([once] * world, BF16)
        for it in range(4):
            if it == 2:                      # new inputs between replays
                parts = data(rng, world, count, BF16)
                for r in range(world):
                    din[r].upload(parts[r])
                once = AO.allreduce_sum(parts, BF16)
                want = AO.allreduce_sum([once] * world, BF16)
            for r in range(world):
                graphs[r].launch()
            for r in range(world):
                streams[r].synchronize()
                assert gpu.lib.atoma_xgmi_status(xs[r]) == 0
                assert np.array_equal(dout[r].numpy(np.uint16, (count,)), want), f"replay {it}, rank {r}"
    finally:
        for x in xs:
            gpu.lib.atoma_xgmi_destroy(x)


def test_missing_peer_times_out_instead_of_hanging(gpu, monkeypatch):
    monkeypatch.setenv("ATOMA_XGMI_TIMEOUT_MS", "300")
    xs = make_ranks(gpu, 2, 1 << 20)
    try:
        x = gpu.DeviceBuffer.zeros((4096,), np.uint16)
        st = gpu.Stream()
        assert gpu.lib.atoma_xgmi_allreduce_sum(xs[0], x.ptr, x.ptr, 4096, BF16, st.s) == 0      # rank 1 never calls
        st.synchronize()
        assert gpu.lib.atoma_xgmi_status(xs[0]) == 2                                              # 1 + the rank that never arrived
        assert gpu.lib.atoma_xgmi_allreduce_sum(xs[0], x.ptr, x.ptr, 4096, BF16, st.s) == -1 and "timed out" in gpu.last_error()
    finally:
        for x_ in xs:
            gpu.lib.atoma_xgmi_destroy(x_)


def test_argument_checks(gpu):
    h = C.c_void_p()
    assert gpu.lib.atoma_xgmi_create(C.byref(h), 0, 9, 0, 1 << 20) == -1 and "world_size" in gpu.last_error()
    assert gpu.lib.atoma_xgmi_create(C.byref(h), 0, 2, 0, 1 << 20) == 0
    x = gpu.DeviceBuffer(4096)
    assert gpu.lib.atoma_xgmi_allreduce_sum(h, x.ptr, x.ptr, 64, BF16, None) == -1 and "connect" in gpu.last_error()
    bad = (C.c_uint8 * 256)()
    assert gpu.lib.atoma_xgmi_connect(h, bad) == -1 and "handle 0" in gpu.last_error()
    gpu.lib.atoma_xgmi_destroy(h)
    one = C.c_void_p()
    assert gpu.lib.atoma_xgmi_create(C.byref(one), 0, 1, 0, 1 << 20) == 0                         # a world of one is a copy
    src = gpu.DeviceBuffer.from_numpy(np.arange(64, dtype=np.uint16))
    assert gpu.lib.atoma_xgmi_allreduce_sum(one, src.ptr, x.ptr, 64, BF16, None) == 0
    assert gpu.lib.atoma_xgmi_allreduce_sum(one, src.ptr, x.ptr, 63, BF16, None) == -1 and "multiple of 16" in gpu.last_error()
    gpu.synchronize()
    assert np.array_equal(x.numpy(np.uint16, (64,)), np.arange(64, dtype=np.uint16))
    gpu.lib.atoma_xgmi_destroy(one)


# ---- two PROCESSES on device 0: the HIP IPC route (what torch.distributed.run / one process per GPU uses) ----
def _ipc_worker(rank, world, conn, device):
    try:
        os.environ["ATOMA_XGMI_TIMEOUT_MS"] = "10000"
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
        import atoma_hip as ah
        from oracle import allreduce_oracle as AO2
        from util import rand_half as rh
        ah.set_device(device)
        h = C.c_void_p()
        cap = (64 << 20) if world == 8 else (2 << 20)
        assert ah.lib.atoma_xgmi_create(C.byref(h), rank, world, device, cap) == 0, ah.last_error()
        one = (C.c_uint8 * 128)()
        assert ah.lib.atoma_xgmi_handle(h, one) == 0, ah.last_error()
        conn.send(bytes(one))
        allb = conn.recv()
        blobs = (C.c_uint8 * len(allb)).from_buffer_copy(allb)
        assert ah.lib.atoma_xgmi_connect(h, blobs) == 0, ah.last_error()
        conn.send("connected")
        assert conn.recv() == "go"
        worst = 0
        # one-shot, the decode message (1 MiB > the one-shot limit -> two-shot), two-shot; the 8 ranks of configs[3] also the prefill chunk's 64 MiB
        for i, count in enumerate((8, 64 * 8192, 700 * 1024) + ((4096 * 8192,) if world == 8 else ())):
            parts = [rh(np.random.default_rng(100 * i + r), (count,), 1) for r in range(world)]
            want = AO2.allreduce_sum(parts, 1)
            dx = ah.DeviceBuffer.from_numpy(parts[rank])
            dy = ah.DeviceBuffer(count * 2)
            for _ in range(3):
                assert ah.lib.atoma_xgmi_allreduce_sum(h, dx.ptr, dy.ptr, count, 1, None) == 0, ah.last_error()
            ah.synchronize()
            assert ah.lib.atoma_xgmi_status(h) == 0, "a wait timed out"
            got = dy.numpy(np.uint16, (count,))
            worst = max(worst, int((got != want).sum()))
        conn.send(("ok", worst))
        assert conn.recv() == "bye"
        ah.lib.atoma_xgmi_destroy(h)
    except Exception as e:      # surface the failure in the parent
        import traceback
        conn.send(("error", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world", [2, 8])
def test_processes_over_hip_ipc(gpu, world):
    """One PROCESS per rank (what torch.distributed.run / bench.py --gpus N starts): staging regions mapped through hipIpcOpenMemHandle.
    world 8 on one device = the whole TP = 8 communicator of configs[3], minus the links."""
    ndev = gpu.lib.atoma_device_count()
    devices = list(range(world)) if ndev >= world else [0] * world
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, pipes[r][1], devices[r])) for r in range(world)]
    for p in procs:
        p.start()
    try:
        def get(r, timeout=300):
            assert pipes[r][0].poll(timeout), f"rank {r} did not answer"
            m = pipes[r][0].recv()
            if isinstance(m, tuple) and m[0] == "error":
                pytest.fail(f"rank {r}: {m[1]}")
            return m
        blobs = b"".join(get(r) for r in range(world))
        for r in range(world):
            pipes[r][0].send(blobs)
        for r in range(world):
            assert get(r) == "connected"
        for r in range(world):
            pipes[r][0].send("go")
        for r in range(world):
            status, mismatches = get(r)
            assert status == "ok" and mismatches == 0, f"rank {r}: {mismatches} elements differ from the rank-order sum"
        for r in range(world):
            pipes[r][0].send("bye")
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()


# ---- >= 2 devices: one rank per device, the communicator route (RCCL bootstrap), direct kernels vs RCCL vs the oracle ----
def _comm_worker(rank, world, conn):
    try:
        os.environ["ATOMA_XGMI_TIMEOUT_MS"] = "10000"
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
        import atoma_hip as ah
        from oracle import allreduce_oracle as AO2
        from util import rand_half as rh
        ah.set_device(rank)
        raw = (C.c_uint8 * 128)()
        if rank == 0:
            assert ah.lib.atoma_comm_unique_id(raw) == 0, ah.last_error()
            conn.send(bytes(raw))
        uid = conn.recv()
        raw = (C.c_uint8 * 128).from_buffer_copy(uid)
        comm = C.c_void_p()
        assert ah.lib.atoma_comm_init(C.byref(comm), rank, world, raw, rank) == 0, ah.last_error()
        info = ah.lib.atoma_comm_info(comm).decode()
        report = {"info": info}
        for count in (64 * 8192, 3 * 1024 * 1024):
            parts = [rh(np.random.default_rng(7 * count + r), (count,), 1) for r in range(world)]
            dx, dy = ah.DeviceBuffer.from_numpy(parts[rank]), ah.DeviceBuffer(count * 2)
            res = {}
            for mode, name in ((0, "rccl"), (1, "xgmi")):
                if mode == 1 and "ready" not in info:
                    continue
                assert ah.lib.atoma_comm_set_mode(comm, mode) == 0, ah.last_error()
                assert ah.lib.atoma_allreduce_sum(comm, dx.ptr, dy.ptr, count, 1, None) == 0, ah.last_error()
                ah.synchronize()
                res[name] = dy.numpy(np.uint16, (count,))
            exact = AO2.allreduce_sum(parts, 1, "exact")
            key = lambda x: np.where(x.astype(np.int32) & 0x8000, 0x8000 - x.astype(np.int32), x.astype(np.int32))
            report[count] = {n: int(np.abs(key(v) - key(exact)).max()) for n, v in res.items()}
            if "xgmi" in res:
                report[count]["xgmi_bit_exact_rank_order"] = bool(np.array_equal(res["xgmi"], AO2.allreduce_sum(parts, 1)))
        conn.send(("ok", report))
        assert conn.recv() == "bye"
        ah.lib.atoma_comm_destroy(comm)
    except Exception as e:
        import traceback
        conn.send(("error", traceback.format_exc() + repr(e)))


def test_rccl_vs_direct_one_rank_per_device(gpu):
    if gpu.lib.atoma_device_count() < 2:
        pytest.skip("needs >= 2 devices (RCCL refuses two ranks on one device); the single-device tests above cover the kernels")
    world = 2
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_comm_worker, args=(r, world, pipes[r][1])) for r in range(world)]
    for p in procs:
        p.start()
    try:
        assert pipes[0][0].poll(120)
        uid = pipes[0][0].recv()
        if isinstance(uid, tuple):
            pytest.fail(uid[1])
        for r in range(world):
            pipes[r][0].send(uid)
        for r in range(world):
            assert pipes[r][0].poll(300), f"rank {r} did not answer"
            status, rep = pipes[r][0].recv()
            assert status == "ok", rep
            assert "ready" in rep["info"], rep["info"]
            for count, d in rep.items():
                if count == "info":
                    continue
                assert d["rccl"] <= 1 and d["xgmi"] <= 1 and d["xgmi_bit_exact_rank_order"], (count, d)
        for r in range(world):
            pipes[r][0].send("bye")
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()


# ---- all-reduce + residual add + RMSNorm in the all-reduce's own launch ----
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("dtype", [BF16, F16])
def test_fused_allreduce_add_rms_norm_equals_the_three_ops(gpu, world, dtype, monkeypatch):
    """atoma_xgmi_allreduce_add_rms_norm against atoma_xgmi_allreduce_sum + atoma_add_rms_norm on the same tensors: bit for bit on every rank, both
    kernels (one-shot / two-shot), the decode message of configs[3] ([64, 8192]), more rows than blocks, a hidden size that is not a multiple
    of a block's 256 vectors, strided outputs; and against the oracle (rank-order sum, add, RMSNorm within one ulp).  Twice in a row and after
    plain all-reduces on the same communicator (sequence numbers and flags are shared)."""
    from oracle import elementwise_oracle as EO
    from oracle import norm_rope_oracle as NR
    monkeypatch.setenv("ATOMA_XGMI_TIMEOUT_MS", "8000")
    monkeypatch.setenv("ATOMA_XGMI_ONESHOT_MAX", str(1 << 20))
    rng = np.random.default_rng(500 + world * 7 + dtype)
    xs = make_ranks(gpu, world, 4 << 20)
    streams = [gpu.Stream() for _ in range(world)]
    eps = 1e-5
    try:
        for rows, hidden, pad in ((64, 8192, 0), (5, 512, 0), (70, 1024, 16), (1, 8, 0), (33, 2056, 8)):
            count = rows * hidden
            parts = [rand_half(rng, (rows, hidden), dtype) for _ in range(world)]
            res = rand_half(rng, (rows, hidden + pad), dtype)
            w = rand_half(rng, (hidden,), dtype)
            din = [gpu.DeviceBuffer.from_numpy(p) for p in parts]
            dres, dw = gpu.DeviceBuffer.from_numpy(res), gpu.DeviceBuffer.from_numpy(w)
            st_x, st_n = hidden + pad, hidden + 2 * pad
            # the three ops, on rank 0's stream after a plain all-reduce of every rank
            dsum = [gpu.DeviceBuffer(count * 2) for _ in range(world)]
            for r in range(world):
                assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], din[r].ptr, dsum[r].ptr, count, dtype, streams[r].s) == 0, gpu.last_error()
            x_ref, n_ref = gpu.DeviceBuffer.zeros((rows, st_x), np.uint16), gpu.DeviceBuffer.zeros((rows, st_n), np.uint16)
            assert gpu.lib.atoma_add_rms_norm(dres.ptr, dsum[0].ptr, dw.ptr, x_ref.ptr, n_ref.ptr, rows, hidden, hidden + pad, hidden, st_x, st_n, eps, dtype, streams[0].s) == 0, gpu.last_error()
            for r in range(world):
                streams[r].synchronize()
            want_x, want_n = x_ref.numpy(np.uint16, (rows, st_x)), n_ref.numpy(np.uint16, (rows, st_n))
            # ... the oracle agrees with them
            s_or = AO.allreduce_sum([p.reshape(-1) for p in parts], dtype).reshape(rows, hidden)
            x_or = EO.add(np.ascontiguousarray(res[:, :hidden]), s_or, dtype)
            assert np.array_equal(want_x[:, :hidden], x_or)
            assert ulps(want_n[:, :hidden], NR.rms_norm(x_or, w, eps, dtype)).max() <= 1
            for mode in (1, 2, 0, 2):
                xo = [gpu.DeviceBuffer.zeros((rows, st_x), np.uint16) for _ in range(world)]
                no = [gpu.DeviceBuffer.zeros((rows, st_n), np.uint16) for _ in range(world)]
                for r in range(world):
                    rc = gpu.lib.atoma_xgmi_allreduce_add_rms_norm(xs[r], din[r].ptr, dres.ptr, dw.ptr, xo[r].ptr, no[r].ptr, rows, hidden, hidden + pad, st_x, st_n,
                                                                   eps, dtype, mode, streams[r].s)
                    assert rc == 0, gpu.last_error()
                for r in range(world):
                    streams[r].synchronize()
                    assert gpu.lib.atoma_xgmi_status(xs[r]) == 0, f"rank {r}: a wait timed out"
                    assert np.array_equal(xo[r].numpy(np.uint16, (rows, st_x)), want_x), f"{rows}x{hidden} mode {mode} rank {r}: residual + sum differs from the three ops"
                    assert np.array_equal(no[r].numpy(np.uint16, (rows, st_n)), want_n), f"{rows}x{hidden} mode {mode} rank {r}: norm differs from the three ops"
        # too large for one launch: refused, not cut
        big = gpu.DeviceBuffer(16)
        assert gpu.lib.atoma_xgmi_allreduce_add_rms_norm(xs[0], big.ptr, big.ptr, big.ptr, big.ptr, big.ptr, 1024, 8192, 8192, 8192, 8192, eps, dtype, 0, None) == -1
        assert "capacity" in gpu.last_error()
    finally:
        for x in xs:
            gpu.lib.atoma_xgmi_destroy(x)


# ---- first-contact failures, injected: the set-up calls of the direct engine have only ever succeeded here ----
@pytest.mark.parametrize("fault,what", [(1, "hipIpcOpenMemHandle"), (2, "hipDeviceEnablePeerAccess"), (3, "hipIpcGetMemHandle")])
def test_injected_setup_failures_are_clean_and_recoverable(gpu, fault, what):
    """atoma_set_option("xgmi_fault", n) makes hipIpcOpenMemHandle / hipDeviceEnablePeerAccess / hipIpcGetMemHandle fail as they might on the
    first multi-GPU box: the call returns -1 with the failing call's name, the handle is NOT connected (an all-reduce on it says so instead
    of touching unmapped memory), nothing has to be torn down by hand -- the same handles connect and sum correctly once the fault is gone,
    and destroy cleanly either way.  (A communicator's `auto` mode answers a failed direct set-up by staying on RCCL: atoma_comm_info says
    "xgmi: unavailable (<the message checked here>)"; that route needs two devices.)"""
    world, count = 2, 4096
    xs = []
    for r in range(world):
        h = C.c_void_p()
        assert gpu.lib.atoma_xgmi_create(C.byref(h), r, world, 0, 1 << 20) == 0, gpu.last_error()
        xs.append(h)
    one = (C.c_uint8 * 128)()
    try:
        assert gpu.lib.atoma_set_option(b"xgmi_fault", fault) == 0
        blobs = (C.c_uint8 * (128 * world))()
        if fault == 3:
            assert gpu.lib.atoma_xgmi_handle(xs[0], one) == -1 and what in gpu.last_error() and "injected" in gpu.last_error()
            gpu.lib.atoma_set_option(b"xgmi_fault", 0)
        for r in range(world):
            assert gpu.lib.atoma_xgmi_handle(xs[r], one) == 0, gpu.last_error()
            C.memmove(C.addressof(blobs) + 128 * r, one, 128)
        if fault == 1:                        # the IPC branch is taken for a peer of ANOTHER process: forge the peer's pid in rank 0's copy
            forged = (C.c_uint8 * (128 * world)).from_buffer_copy(bytes(blobs))
            C.memmove(C.addressof(forged) + 128 * 1 + 16, (C.c_int64 * 1)(os.getpid() + 1), 8)
            assert gpu.lib.atoma_xgmi_connect(xs[0], forged) == -1 and what in gpu.last_error() and "injected" in gpu.last_error()
        if fault == 2:
            assert gpu.lib.atoma_xgmi_connect(xs[0], blobs) == -1 and what in gpu.last_error() and "injected" in gpu.last_error()
        if fault in (1, 2):                   # not connected: loud, no launch
            x = gpu.DeviceBuffer.zeros((count,), np.uint16)
            assert gpu.lib.atoma_xgmi_allreduce_sum(xs[0], x.ptr, x.ptr, count, BF16, None) == -1 and "atoma_xgmi_connect" in gpu.last_error()
    finally:
        gpu.lib.atoma_set_option(b"xgmi_fault", 0)
    for r in range(world):                    # the fault is gone: the SAME handles come up and work
        assert gpu.lib.atoma_xgmi_connect(xs[r], blobs) == 0, gpu.last_error()
    rng = np.random.default_rng(fault)
    parts = [rand_half(rng, (count,), BF16) for _ in range(world)]
    want = AO.allreduce_sum(parts, BF16)
    streams = [gpu.Stream() for _ in range(world)]
    ins = [gpu.DeviceBuffer.from_numpy(p_) for p_ in parts]
    outs = [gpu.DeviceBuffer(count * 2) for _ in range(world)]
    for r in range(world):
        assert gpu.lib.atoma_xgmi_allreduce_sum(xs[r], ins[r].ptr, outs[r].ptr, count, BF16, streams[r].s) == 0, gpu.last_error()
    for s_ in streams:
        s_.synchronize()
    for r in range(world):
        assert np.array_equal(outs[r].numpy(np.uint16, (count,)), want) and gpu.lib.atoma_xgmi_status(xs[r]) == 0
        gpu.lib.atoma_xgmi_destroy(xs[r])
