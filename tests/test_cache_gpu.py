"""Bit-exact parity of the KV-cache kernels with the oracle, through the C ABI (reference tests:
/root/reference/csrc/tests/cache_manager_tests.rs)."""
import ctypes as C

import numpy as np
import pytest

from oracle import cache_oracle as CO
from oracle.halfs import F16, BF16
from util import rand_half

pytestmark = pytest.mark.gpu


def _reshape(gpu, key, val, kc, vc, slots, dtype, key_stride=None, val_stride=None, key_buf=None):
    T, hk, d = key.shape
    nb, page = kc.shape[:2]
    dk = gpu.DeviceBuffer.from_numpy(key_buf if key_buf is not None else key)
    dv, dkc, dvc = (gpu.DeviceBuffer.from_numpy(a) for a in (val, kc, vc))
    ds = gpu.DeviceBuffer.from_numpy(np.asarray(slots, np.int64))
    gpu.lib.reshape_and_cache_flash(dk.ptr, dv.ptr, dkc.ptr, dvc.ptr, ds.ptr, page * hk * d, T, hk, d, page,
                                    key_stride or hk * d, val_stride or hk * d, dtype, None)
    gpu.check()
    gpu.synchronize()
    return dkc.numpy(), dvc.numpy()


@pytest.mark.parametrize("dtype", [F16, BF16])
def test_reshape_and_cache_flash_reference_case(gpu, dtype):
    """cache_manager_tests.rs:553-616: 10 tokens -> slots 0..9 over 2 pages of 8."""
    rng = np.random.default_rng(1)
    key, val = rand_half(rng, (10, 4, 64), dtype), rand_half(rng, (10, 4, 64), dtype)
    kc, vc = np.zeros((2, 8, 4, 64), np.uint16), np.zeros((2, 8, 4, 64), np.uint16)
    gk, gv = _reshape(gpu, key, val, kc, vc, np.arange(10), dtype)
    CO.reshape_and_cache_flash(key, val, kc, vc, np.arange(10))
    assert np.array_equal(gk, kc) and np.array_equal(gv, vc)
    for i in range(10):
        assert np.array_equal(gk[i // 8, i % 8], key[i]) and np.array_equal(gv[i // 8, i % 8], val[i])


def test_reshape_and_cache_flash_golden_fixture(gpu):
    c = np.load(__file__.replace("test_cache_gpu.py", "golden/oracle_cases.npz"))
    z = np.zeros_like(c["r1_kcache"])
    gk, gv = _reshape(gpu, c["r1_key"], c["r1_val"], z, z.copy(), np.arange(10), F16)
    assert np.array_equal(gk, c["r1_kcache"]) and np.array_equal(gv, c["r1_vcache"])


@pytest.mark.parametrize("T,hk,d,page,nb", [(256, 8, 128, 16, 40), (2048, 8, 128, 16, 160), (37, 2, 64, 32, 4),
                                            (5, 1, 128, 16, 2), (33, 3, 8, 16, 4)])
def test_reshape_and_cache_flash_random_slots_and_padding(gpu, T, hk, d, page, nb):
    rng = np.random.default_rng(T)
    key, val = rand_half(rng, (T, hk, d), BF16), rand_half(rng, (T, hk, d), BF16)
    kc, vc = rand_half(rng, (nb, page, hk, d), BF16), rand_half(rng, (nb, page, hk, d), BF16)
    slots = rng.permutation(nb * page)[:T].astype(np.int64)
    slots[rng.integers(0, T, max(1, T // 8))] = -1           # padding tokens are skipped
    gk, gv = _reshape(gpu, key, val, kc, vc, slots, BF16)
    CO.reshape_and_cache_flash(key, val, kc, vc, slots)
    assert np.array_equal(gk, kc) and np.array_equal(gv, vc)


def test_reshape_and_cache_flash_strided_and_unaligned_sources(gpu):
    """key.stride(0) != hk*d (a q/k/v slice of a fused projection) and a row length that breaks
    16-byte alignment (scalar fallback path)."""
    rng = np.random.default_rng(5)
    T, hk, d, page, nb = 19, 2, 64, 16, 3
    fused = rand_half(rng, (T, 3 * hk * d), F16)
    key = fused[:, hk * d: 2 * hk * d].reshape(T, hk, d)
    val = rand_half(rng, (T, hk, d), F16)
    kc, vc = np.zeros((nb, page, hk, d), np.uint16), np.zeros((nb, page, hk, d), np.uint16)
    slots = rng.permutation(nb * page)[:T].astype(np.int64)
    dk = gpu.DeviceBuffer.from_numpy(fused)
    dv, dkc, dvc = (gpu.DeviceBuffer.from_numpy(a) for a in (val, kc, vc))
    ds = gpu.DeviceBuffer.from_numpy(slots)
    gpu.lib.reshape_and_cache_flash(dk.ptr + hk * d * 2, dv.ptr, dkc.ptr, dvc.ptr, ds.ptr, page * hk * d, T, hk, d,
                                    page, 3 * hk * d, hk * d, F16, None)
    gpu.check(); gpu.synchronize()
    CO.reshape_and_cache_flash(key, val, kc, vc, slots)
    assert np.array_equal(dkc.numpy(), kc) and np.array_equal(dvc.numpy(), vc)
    # odd head size 4 -> rows of 8 bytes: vector path impossible
    T, hk, d = 7, 1, 4
    key, val = rand_half(rng, (T, hk, d), F16), rand_half(rng, (T, hk, d), F16)
    kc, vc = np.zeros((1, 16, hk, d), np.uint16), np.zeros((1, 16, hk, d), np.uint16)
    gk, gv = _reshape(gpu, key, val, kc, vc, np.arange(T)[::-1].copy(), F16)
    CO.reshape_and_cache_flash(key, val, kc, vc, np.arange(T)[::-1])
    assert np.array_equal(gk, kc) and np.array_equal(gv, vc)


def test_reshape_and_cache_flash_rejects_unknown_dtype(gpu):
    gpu.lib.reshape_and_cache_flash(None, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 2, None)
    assert "dtype" in gpu.last_error()
    gpu.lib.atoma_clear_error()


def _copy_blocks(gpu, ks, vs, mapping, fn):
    dks = [gpu.DeviceBuffer.from_numpy(k) for k in ks]
    dvs = [gpu.DeviceBuffer.from_numpy(v) for v in vs]
    kp = gpu.DeviceBuffer.from_numpy(np.array([b.ptr for b in dks], np.int64))
    vp = gpu.DeviceBuffer.from_numpy(np.array([b.ptr for b in dvs], np.int64))
    mp = gpu.DeviceBuffer.from_numpy(np.asarray(mapping, np.int64))
    numel = int(np.prod(ks[0].shape[1:]))
    getattr(gpu.lib, fn)(kp.ptr, vp.ptr, mp.ptr, len(ks), len(mapping), numel, None)
    gpu.check(); gpu.synchronize()
    return [b.numpy() for b in dks], [b.numpy() for b in dvs]


@pytest.mark.parametrize("fn,dtype", [("copy_blocks_f16", F16), ("copy_blocks_bf16", BF16)])
def test_copy_blocks_reference_case(gpu, fn, dtype):
    """cache_manager_tests.rs:242-347: 2 layers, [4, 64, 2, 8], mapping [[0,2],[1,3]]."""
    rng = np.random.default_rng(2)
    ks = [rand_half(rng, (4, 64, 2, 8), dtype) for _ in range(2)]
    vs = [rand_half(rng, (4, 64, 2, 8), dtype) for _ in range(2)]
    gk, gv = _copy_blocks(gpu, ks, vs, [[0, 2], [1, 3]], fn)
    CO.copy_blocks(ks, vs, [[0, 2], [1, 3]])
    for l in range(2):
        assert np.array_equal(gk[l], ks[l]) and np.array_equal(gv[l], vs[l])
        assert np.array_equal(gk[l][2], gk[l][0]) and np.array_equal(gk[l][3], gk[l][1])


@pytest.mark.parametrize("L,P,shape", [(32, 64, (160, 16, 8, 128)), (3, 1, (5, 16, 8, 128)), (2, 7, (16, 16, 1, 12))])
def test_copy_blocks_llama_pages_and_odd_sizes(gpu, L, P, shape):
    rng = np.random.default_rng(L * 100 + P)
    ks = [rand_half(rng, shape, BF16) for _ in range(L)]
    vs = [rand_half(rng, shape, BF16) for _ in range(L)]
    perm = rng.permutation(shape[0])
    mapping = np.stack([perm[:P], perm[P:2 * P]], 1)          # distinct srcs and dsts
    gk, gv = _copy_blocks(gpu, ks, vs, mapping, "copy_blocks_bf16")
    CO.copy_blocks(ks, vs, mapping)
    for l in range(L):
        assert np.array_equal(gk[l], ks[l]) and np.array_equal(gv[l], vs[l])


def _swap(gpu, src, dst, mapping, kind, src_ptr, dst_ptr):
    m = np.asarray(sorted(mapping.items()), np.int64)
    page_bytes = int(np.prod(src.shape[1:])) * 2
    rc = gpu.lib.atoma_swap_blocks(src_ptr, dst_ptr, m.ctypes.data, len(m), page_bytes, kind, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()


@pytest.mark.parametrize("dtype,mapping", [(F16, {0: 2, 1: 0}), (BF16, {0: 1, 2: 0})])
def test_swap_blocks_gpu_to_gpu_reference_case(gpu, dtype, mapping):
    """cache_manager_tests.rs:64-102."""
    rng = np.random.default_rng(3)
    src, dst = rand_half(rng, (3, 16, 2, 8), dtype), rand_half(rng, (3, 16, 2, 8), dtype)
    ds, dd = gpu.DeviceBuffer.from_numpy(src), gpu.DeviceBuffer.from_numpy(dst)
    _swap(gpu, src, dst, mapping, 0, ds.ptr, dd.ptr)
    CO.swap_blocks(src, dst, mapping)
    assert np.array_equal(dd.numpy(), dst) and np.array_equal(ds.numpy(), src)


@pytest.mark.parametrize("pinned", [False, True, "registered"])
def test_swap_blocks_cpu_gpu_round_trip(gpu, pinned):
    """cache_manager_tests.rs:104-186 both directions: pageable host memory (the library's pinned bounce ring), atoma_host_alloc and
    a caller-owned allocation pinned in place with atoma_host_register (one gather/scatter kernel over PCIe); swap-out then swap-in
    restores the pages."""
    rng = np.random.default_rng(4)
    shape = (40, 16, 8, 128)                                   # 32 KiB pages (Llama-3.1-8B)
    nbytes = int(np.prod(shape)) * 2
    gpu_cache = rand_half(rng, shape, F16)
    dg = gpu.DeviceBuffer.from_numpy(gpu_cache)
    if pinned is True:
        hptr = gpu.lib.atoma_host_alloc(nbytes)
        assert hptr
        host = np.ctypeslib.as_array(C.cast(hptr, C.POINTER(C.c_uint16)), shape=(nbytes // 2,)).reshape(shape)
    else:
        host = np.empty(shape, np.uint16)
        hptr = host.ctypes.data
        if pinned == "registered":
            assert gpu.lib.atoma_host_register(hptr, nbytes) == 0, gpu.last_error()
    host[...] = rand_half(rng, shape, F16)
    want_host = host.copy()
    out_map = {int(s): int(d) for s, d in zip(rng.permutation(40)[:17], rng.permutation(40)[:17])}
    _swap(gpu, gpu_cache, host, out_map, 2, dg.ptr, hptr)        # gpu -> cpu
    CO.swap_blocks(gpu_cache, want_host, out_map)
    assert np.array_equal(host, want_host)
    in_map = {d: s for s, d in out_map.items()}
    dg.fill_bytes(0)
    _swap(gpu, host, gpu_cache, in_map, 1, hptr, dg.ptr)         # cpu -> gpu
    back = dg.numpy()
    for s in out_map:
        assert np.array_equal(back[s], gpu_cache[s])
    untouched = [i for i in range(40) if i not in out_map]
    assert not back[untouched].any()
    if pinned is True:
        gpu.lib.atoma_host_free(hptr)
    elif pinned == "registered":
        assert gpu.lib.atoma_host_unregister(hptr) == 0, gpu.last_error()


def test_swap_blocks_multi_all_layers(gpu):
    rng = np.random.default_rng(6)
    L, shape = 6, (12, 16, 2, 64)
    srcs = [rand_half(rng, shape, BF16) for _ in range(2 * L)]
    dsts = [rand_half(rng, shape, BF16) for _ in range(2 * L)]
    ds = [gpu.DeviceBuffer.from_numpy(a) for a in srcs]
    dd = [gpu.DeviceBuffer.from_numpy(a) for a in dsts]
    mapping = {1: 0, 5: 3, 7: 11}
    m = np.asarray(sorted(mapping.items()), np.int64)
    sp = (C.c_void_p * len(ds))(*[b.ptr for b in ds])
    dp = (C.c_void_p * len(dd))(*[b.ptr for b in dd])
    rc = gpu.lib.atoma_swap_blocks_multi(sp, dp, len(ds), m.ctypes.data, len(m), int(np.prod(shape[1:])) * 2, 0, None)
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    for i in range(2 * L):
        CO.swap_blocks(srcs[i], dsts[i], mapping)
        assert np.array_equal(dd[i].numpy(), dsts[i])


@pytest.mark.parametrize("pinned", [True, False])
def test_swap_blocks_multi_cpu_gpu_config4_shape(gpu, pinned):
    """BASELINE configs[4]: every layer's K and V of Llama-3.1-8B fp16 -- 64 tensors -- 256 random distinct
    pages each way, through atoma_swap_blocks_multi kinds 2 (gpu -> cpu) and 1 (cpu -> gpu), pinned (one gather/scatter
    kernel over PCIe) and pageable (per-page memcpy).  Bit-exact against the oracle per tensor, untouched pages stay
    untouched, and swap-out followed by swap-in restores the GPU cache (worker.rs:602-632 loops the layers)."""
    rng = np.random.default_rng(40 + pinned)
    n_t, nb, npairs = 64, 288, 256
    shape = (nb, 16, 8, 128)                                    # 32 KiB pages
    page_bytes = int(np.prod(shape[1:])) * 2
    nbytes = nb * page_bytes
    base = rand_half(rng, (n_t // 8,) + shape, F16)             # 8 distinct tensors' worth of data, re-used with a per-tensor tag
    gpu_np = [np.ascontiguousarray(base[i % len(base)] ^ np.uint16(i)) for i in range(n_t)]
    dgs = [gpu.DeviceBuffer.from_numpy(a) for a in gpu_np]
    hosts, hptrs = [], []
    for i in range(n_t):
        if pinned:
            p = gpu.lib.atoma_host_alloc(nbytes)
            assert p
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(nbytes // 2,)).reshape(shape)
        else:
            a = np.empty(shape, np.uint16)
            p = a.ctypes.data
        a[...] = np.uint16(0x1111 + i)                          # recognisable background
        hosts.append(a)
        hptrs.append(p)
    src_pages, dst_pages = rng.permutation(nb)[:npairs], rng.permutation(nb)[:npairs]
    out_map = np.stack([src_pages, dst_pages], 1).astype(np.int64)
    gp = (C.c_void_p * n_t)(*[b.ptr for b in dgs])
    hp = (C.c_void_p * n_t)(*hptrs)
    rc = gpu.lib.atoma_swap_blocks_multi(gp, hp, n_t, out_map.ctypes.data, npairs, page_bytes, 2, None)     # gpu -> cpu
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    mapping = {int(s): int(d) for s, d in out_map}
    for i in range(n_t):
        want = np.full(shape, np.uint16(0x1111 + i), np.uint16)
        CO.swap_blocks(gpu_np[i], want, mapping)
        assert np.array_equal(hosts[i], want), f"gpu->cpu tensor {i}"
    # wipe the GPU side, swap back in through the inverse map
    for b in dgs:
        b.fill_bytes(0)
    in_map = np.ascontiguousarray(out_map[:, ::-1])
    rc = gpu.lib.atoma_swap_blocks_multi(hp, gp, n_t, in_map.ctypes.data, npairs, page_bytes, 1, None)      # cpu -> gpu
    assert rc == 0, gpu.last_error()
    gpu.synchronize()
    touched = np.zeros(nb, bool)
    touched[src_pages] = True
    for i in range(n_t):
        back = dgs[i].numpy(np.uint16, shape)
        assert np.array_equal(back[touched], gpu_np[i][touched]), f"cpu->gpu tensor {i}"
        assert not back[~touched].any(), f"cpu->gpu tensor {i}: a page outside the map was written"
    if pinned:
        for p in hptrs:
            gpu.lib.atoma_host_free(p)
