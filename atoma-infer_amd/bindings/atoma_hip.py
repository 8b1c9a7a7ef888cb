"""ctypes binding of libatoma_hip.so + a minimal HIP-runtime device-buffer helper.

This is plumbing for tests/, bench.py and __graft_entry__.py -- the product is the C ABI
(include/atoma_hip.h).  No torch: device memory, events and streams come straight from
libamdhip64.  Importing this module fails loudly if the HIP extension is not built.
"""
import ctypes as C
import os

# The host driver of the MI355X pool only supports dmabuf IPC: without this hipIpcGetMemHandle (the direct all-reduce's handle exchange, RCCL's own
# peer set-up) fails with "invalid argument".  It must be in the environment before the HIP runtime initialises; a value the caller set is kept.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ATOMA_HIP_LIB: another build of the SAME library (the probe variants of `make timing` / `make fp8p`); never a different backend
LIB_PATH = os.environ.get("ATOMA_HIP_LIB") or os.path.join(_HERE, "..", "lib", "libatoma_hip.so")

F16, BF16 = 0, 1

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C atoma-infer_amd` (or python -c "
        "'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")


class _Lib(C.CDLL):
    """With ATOMA_HIP_LIB pointing at an OLDER build of the library (tools/ab_binary.py: the previous binary beside the new one), an
    entry point that build lacks fails when it is CALLED, not when this module declares its argument types.  Without the override a
    missing symbol is an AttributeError at import, as before."""

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if not os.environ.get("ATOMA_HIP_LIB") or name.startswith("__"):
                raise

            def missing(*a, **k):
                raise RuntimeError(f"{name} is not exported by {LIB_PATH}")
            setattr(self, name, missing)
            return missing


lib = _Lib(os.path.abspath(LIB_PATH))

_vp, _i32p, _i64p = C.c_void_p, C.c_void_p, C.c_void_p
_u32, _i64, _f32, _int, _bool = C.c_uint32, C.c_int64, C.c_float, C.c_int, C.c_bool

# ---- section 1: reference FFI (csrc/src/ffi.rs) ---------------------------------------------
_RUN_MHA_ARGS = [
    _vp, _vp, _vp, _vp, _vp, _vp,          # q k v o softmax_lse alibi_slopes
    _i32p, _i32p,                          # cu_seqlens_q, cu_seqlens_k
    _bool,                                 # is_seqlens_k_cumulative
    _u32, _u32, _u32, _u32, _u32,          # q/k/v/o/alibi batch strides
    _u32, _u32, _u32, _u32,                # row strides
    _u32, _u32, _u32, _u32,                # head strides
    _u32,                                  # num_splits
    _u32, _u32, _u32, _u32, _u32,          # b h h_k d d_rounded
    _f32, _f32,                            # softmax_scale, scale_softmax_log2
    _i32p, _u32, _int,                     # block_table, stride, page_block_size
    _i32p,                                 # seqused_k
    _u32, _u32, _u32, _u32,                # seqlen_q, seqlen_k, rounded x2
    _int, _int,                            # is_bf16, is_causal
    _int, _int,                            # window left/right
    _f32, _bool, _bool,                    # softcap, unpadded_lse, force_split_kernel
    _vp, _vp,                              # lseaccum, oaccum
]
lib.run_mha.argtypes = _RUN_MHA_ARGS
lib.run_mha.restype = None
lib.run_mha_stream.argtypes = _RUN_MHA_ARGS + [_vp]
lib.run_mha_stream.restype = None
for _n in ("copy_blocks_f16", "copy_blocks_bf16"):
    getattr(lib, _n).argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _vp]
    getattr(lib, _n).restype = None
lib.reshape_and_cache_flash.argtypes = [_vp, _vp, _vp, _vp, _i64p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _u32, _vp]
lib.reshape_and_cache_flash.restype = None

# ---- section 2 ---------------------------------------------------------------------------------
lib.atoma_last_error.restype = C.c_char_p
lib.atoma_last_error.argtypes = []
lib.atoma_clear_error.restype = None
lib.atoma_num_splits_heuristic.argtypes = [_i64, _i64, _i64, _i64]
lib.atoma_num_splits_heuristic.restype = _int
lib.atoma_compute_num_splits.argtypes = [_i64, _i64, _i64, _i64, _i64, _int]
lib.atoma_compute_num_splits.restype = _int
lib.atoma_swap_blocks.argtypes = [_vp, _vp, _i64p, _i64, _i64, _int, _vp]
lib.atoma_swap_blocks.restype = _int
lib.atoma_swap_blocks_multi.argtypes = [_vp, _vp, _i64, _i64p, _i64, _i64, _int, _vp]
lib.atoma_swap_blocks_multi.restype = _int
lib.atoma_host_alloc.argtypes = [C.c_size_t]
lib.atoma_host_alloc.restype = _vp
lib.atoma_host_free.argtypes = [_vp]
lib.atoma_host_free.restype = None
lib.atoma_host_register.argtypes = [_vp, C.c_size_t]
lib.atoma_host_register.restype = C.c_int
lib.atoma_host_unregister.argtypes = [_vp]
lib.atoma_host_unregister.restype = C.c_int
lib.atoma_device_count.restype = _int
lib.atoma_set_option.argtypes = [C.c_char_p, _int]
lib.atoma_set_option.restype = _int
lib.atoma_num_cus.argtypes = [_int]
lib.atoma_num_cus.restype = _int


def _opt(name, argtypes, restype=_int):
    """Entry points added later in the round: bind when present."""
    if hasattr(lib, name):
        f = getattr(lib, name)
        f.argtypes, f.restype = argtypes, restype
        return f
    return None


_opt("atoma_xgmi_allreduce_add_rms_norm", [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _f32, _int, _int, _vp])
_opt("atoma_allreduce_add_rms_norm", [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _int, _vp])
_opt("atoma_warmup_prefill", [_vp, _i64, _i64, _i64])
_opt("atoma_debug_sync_words", [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)])
_opt("atoma_debug_workspace", [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)])
_opt("atoma_hint_decode_lengths", [_i64, _i64, _i64])
_opt("atoma_debug_launch_epoch", [_vp, _vp])
_opt("atoma_rms_norm", [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _int, _vp])
_opt("atoma_add_rms_norm", [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _int, _vp])
_opt("atoma_rope", [_vp, _vp, _vp, _vp, _i64p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp])
_opt("atoma_rope_qk", [_vp, _vp, _vp, _vp, _i64p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp])
_opt("atoma_rope_qk_cache", [_vp, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _i64p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                             _int, _int, _vp])
_opt("atoma_linear_decode_qkv_rope_cache", [_vp, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _i64p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp])
_opt("atoma_embedding", [_vp, _int, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_add", [_vp, _vp, _vp, _i64, _int, _vp])
_opt("atoma_silu_mul", [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_linear_decode_residual", [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_linear_decode_silu_mul", [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_linear_decode_rmsnorm", [_vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_linear_decode_rmsnorm_silu_mul", [_vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_linear_decode", [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_linear", [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_topk_rows", [_vp, _i64, _i64, _i64, _int, _i64, _vp, _vp, _vp])
_opt("atoma_sample_rows", [_vp, _i64, _i64, _i64, _int, _f32, _i64, _f32, _vp, _vp, _vp, _vp])
_opt("atoma_argmax_rows", [_vp, _i64, _i64, _i64, _int, _vp, _vp, _vp])
_opt("atoma_rope_table", [_vp, _vp, _i64, _i64, _f32, _f32, _f32, _f32, _i64, _int])
_opt("atoma_reshape_and_cache_flash_fp8", [_vp, _vp, _vp, _vp, _i64p, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp])
_opt("atoma_rope_qk_cache_fp8", [_vp, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _vp, _vp, _i64p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp])
_opt("atoma_paged_decode_fp8", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _int, _vp])
_opt("atoma_kv_blocks_packed_size", [_i64, _i64, _i64, _i64, _i64, _int], _i64)
_opt("atoma_kv_pack_blocks", [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _int, _vp, _vp, _vp, _i64, _vp])
_opt("atoma_kv_read_header", [_vp, _i64, _vp])
_opt("atoma_kv_unpack_blocks", [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp, _vp, _vp])
_opt("atoma_last_decode_kernel", [], C.c_char_p)
_opt("atoma_warmup", [_vp, _i64, _i64, _i64, _i64, _i64, _i64])
_opt("atoma_reserve_workspace", [_vp, _i64])
_opt("atoma_release_workspaces", [])
_opt("atoma_reset_sync_counters", [_vp])
_opt("atoma_comm_unique_id", [_vp])
_opt("atoma_comm_init", [C.POINTER(_vp), _int, _int, _vp, _int])
_opt("atoma_allreduce_sum", [_vp, _vp, _vp, _i64, _int, _vp])
_opt("atoma_comm_destroy", [_vp])
_opt("atoma_comm_set_mode", [_vp, _int])
_opt("atoma_comm_info", [_vp], C.c_char_p)
_opt("atoma_xgmi_create", [C.POINTER(_vp), _int, _int, _int, _i64])
_opt("atoma_xgmi_handle", [_vp, _vp])
_opt("atoma_xgmi_connect", [_vp, _vp])
_opt("atoma_xgmi_allreduce_sum", [_vp, _vp, _vp, _i64, _int, _vp])
_opt("atoma_xgmi_allreduce_sum_mode", [_vp, _vp, _vp, _i64, _int, _int, _vp])
_opt("atoma_xgmi_status", [_vp])
_opt("atoma_xgmi_capacity", [_vp], _i64)
_opt("atoma_xgmi_destroy", [_vp])


def last_error():
    return lib.atoma_last_error().decode()


def check():
    e = last_error()
    if e:
        raise RuntimeError(e)


# ---- HIP runtime (device memory / events) ------------------------------------------------------
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(_vp), C.c_size_t]
hip.hipFree.argtypes = [_vp]
hip.hipMemcpy.argtypes = [_vp, _vp, C.c_size_t, _int]
hip.hipMemset.argtypes = [_vp, _int, C.c_size_t]
hip.hipMemcpyAsync.argtypes = [_vp, _vp, C.c_size_t, _int, _vp]
hip.hipMemsetAsync.argtypes = [_vp, _int, C.c_size_t, _vp]
hip.hipEventCreate.argtypes = [C.POINTER(_vp)]
hip.hipEventRecord.argtypes = [_vp, _vp]
hip.hipEventSynchronize.argtypes = [_vp]
hip.hipEventElapsedTime.argtypes = [C.POINTER(_f32), _vp, _vp]
hip.hipEventDestroy.argtypes = [_vp]
hip.hipStreamCreate.argtypes = [C.POINTER(_vp)]
hip.hipStreamSynchronize.argtypes = [_vp]
hip.hipStreamDestroy.argtypes = [_vp]
hip.hipSetDevice.argtypes = [_int]
hip.hipStreamBeginCapture.argtypes = [_vp, _int]
hip.hipStreamEndCapture.argtypes = [_vp, C.POINTER(_vp)]
hip.hipGraphInstantiate.argtypes = [C.POINTER(_vp), _vp, _vp, _vp, C.c_size_t]
hip.hipGraphLaunch.argtypes = [_vp, _vp]
hip.hipGraphExecDestroy.argtypes = [_vp]
hip.hipGraphDestroy.argtypes = [_vp]
hip.hipGetErrorString.restype = C.c_char_p
hip.hipGetErrorString.argtypes = [_int]
H2D, D2H, D2D = 1, 2, 3


def hip_check(code, what="hip"):
    if code != 0:
        raise RuntimeError(f"{what}: {hip.hipGetErrorString(code).decode()} ({code})")


def set_device(i):
    hip_check(hip.hipSetDevice(i), "hipSetDevice")


def synchronize():
    hip_check(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


class DeviceBuffer:
    """A hipMalloc'ed region; `.ptr` is the raw device address."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = _vp()
        hip_check(hip.hipMalloc(C.byref(p), max(self.nbytes, 16)), f"hipMalloc({self.nbytes})")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        if a.nbytes:
            hip_check(hip.hipMemcpy(buf.ptr, a.ctypes.data, a.nbytes, H2D), "hipMemcpy H2D")
        buf.shape, buf.dtype = a.shape, a.dtype
        return buf

    @classmethod
    def zeros(cls, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = cls(n)
        if n:
            hip_check(hip.hipMemset(buf.ptr, 0, n), "hipMemset")
        buf.shape, buf.dtype = tuple(shape), np.dtype(dtype)
        return buf

    def fill_bytes(self, value):
        hip_check(hip.hipMemset(self.ptr, value, self.nbytes), "hipMemset")

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        hip_check(hip.hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, H2D), "hipMemcpy H2D")

    def numpy(self, dtype=None, shape=None):
        dtype = np.dtype(dtype if dtype is not None else self.dtype)
        shape = tuple(shape if shape is not None else self.shape)
        out = np.empty(shape, dtype)
        if out.nbytes:
            hip_check(hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, D2H), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            hip.hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self):
        e = _vp()
        hip_check(hip.hipEventCreate(C.byref(e)), "hipEventCreate")
        self.e = e.value

    def record(self, stream=None):
        hip_check(hip.hipEventRecord(self.e, stream), "hipEventRecord")

    def synchronize(self):
        hip_check(hip.hipEventSynchronize(self.e), "hipEventSynchronize")

    def elapsed_ms(self, end):
        ms = _f32()
        hip_check(hip.hipEventElapsedTime(C.byref(ms), self.e, end.e), "hipEventElapsedTime")
        return ms.value


class Stream:
    def __init__(self):
        s = _vp()
        hip_check(hip.hipStreamCreate(C.byref(s)), "hipStreamCreate")
        self.s = s

    def synchronize(self):
        hip_check(hip.hipStreamSynchronize(self.s), "hipStreamSynchronize")

    def __del__(self):
        if getattr(self, "s", None):
            hip.hipStreamDestroy(self.s)
            self.s = None


class Graph:
    """hipGraph captured from a stream: `with Graph.capture(stream) as g: <launches on stream>`, then g.launch()."""

    def __init__(self, stream):
        self.stream, self.graph, self.exe = stream, _vp(), _vp()

    @classmethod
    def capture(cls, stream):
        return cls(stream)

    def __enter__(self):
        hip_check(hip.hipStreamBeginCapture(self.stream.s, 0), "hipStreamBeginCapture")   # hipStreamCaptureModeGlobal
        return self

    def __exit__(self, et, ev, tb):
        rc = hip.hipStreamEndCapture(self.stream.s, C.byref(self.graph))
        if et is None:
            hip_check(rc, "hipStreamEndCapture")
            hip_check(hip.hipGraphInstantiate(C.byref(self.exe), self.graph, None, None, 0), "hipGraphInstantiate")
        return False

    def launch(self):
        hip_check(hip.hipGraphLaunch(self.exe, self.stream.s), "hipGraphLaunch")

    def __del__(self):
        if getattr(self, "exe", None):
            hip.hipGraphExecDestroy(self.exe)
        if getattr(self, "graph", None):
            hip.hipGraphDestroy(self.graph)


def _ptr(x):
    if x is None:
        return None
    return x.ptr if isinstance(x, DeviceBuffer) else x


LOG2E = 1.4426950408889634


def run_mha(q, k, v, o, *, b, h, h_k, d, seqlen_q, seqlen_k, softmax_scale, is_bf16,
            q_strides, k_strides, v_strides, o_strides, is_causal=0, cu_seqlens_q=None,
            cu_seqlens_k=None, is_seqlens_k_cumulative=True, block_table=None,
            block_table_batch_stride=0, page_block_size=0, seqused_k=None, alibi_slopes=None,
            alibi_slopes_batch_stride=0, softmax_lse=None, num_splits=0, force_split_kernel=False,
            unpadded_lse=True, window=(-1, -1), softcap=0.0, lseaccum=None, oaccum=None, stream=None,
            use_stream_entry=False):
    """Thin keyword wrapper over the 47-argument reference entry point.  *_strides are
    (batch, row, head) in elements."""
    rnd = lambda x, m: (x + m - 1) // m * m
    args = [_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(softmax_lse), _ptr(alibi_slopes),
            _ptr(cu_seqlens_q), _ptr(cu_seqlens_k), bool(is_seqlens_k_cumulative),
            q_strides[0], k_strides[0], v_strides[0], o_strides[0], alibi_slopes_batch_stride,
            q_strides[1], k_strides[1], v_strides[1], o_strides[1],
            q_strides[2], k_strides[2], v_strides[2], o_strides[2],
            num_splits, b, h, h_k, d, rnd(d, 32), float(softmax_scale), float(softmax_scale * LOG2E),
            _ptr(block_table), block_table_batch_stride, page_block_size, _ptr(seqused_k),
            seqlen_q, seqlen_k, rnd(seqlen_q, 128), rnd(seqlen_k, 128), int(is_bf16), int(is_causal),
            window[0], window[1], float(softcap), bool(unpadded_lse), bool(force_split_kernel),
            _ptr(lseaccum), _ptr(oaccum)]
    if use_stream_entry or stream is not None:
        lib.run_mha_stream(*args, stream)
    else:
        lib.run_mha(*args)
    check()


# ---- section 3: host mirror of the reference's Candle operator layer ---------------------------
F32, U32, I64, U8, I32 = 2, 3, 4, 5, 6
_DT_SIZE = {F16: 2, BF16: 2, F32: 4, U32: 4, I64: 8, U8: 1, I32: 4}


class Tensor(C.Structure):
    """include/atoma_hip.h `atoma_tensor`: a Candle tensor as the ops see it."""
    _fields_ = [("data", _vp), ("dtype", C.c_int32), ("device", C.c_int32), ("rank", C.c_int32), ("_pad", C.c_int32),
                ("shape", _i64 * 5), ("stride", _i64 * 5)]


def tensor(data, shape, dtype, device=0, strides=None, offset_elems=0):
    """Descriptor over a DeviceBuffer / raw pointer / numpy array (device=-1)."""
    t = Tensor()
    if isinstance(data, np.ndarray):
        ptr, device = data.ctypes.data, -1
    else:
        ptr = _ptr(data)
    t.data = (ptr or 0) + offset_elems * _DT_SIZE[dtype] if ptr is not None else None
    t.dtype, t.device, t.rank = dtype, device, len(shape)
    if strides is None:
        strides, acc = [], 1
        for s in reversed(shape):
            strides.insert(0, acc)
            acc *= s
    for i, (s, st) in enumerate(zip(shape, strides)):
        t.shape[i], t.stride[i] = s, st
    return t


class AttnMetadata(C.Structure):
    _fields_ = [("slot_mapping", C.POINTER(Tensor)), ("num_prefill_tokens", _i64), ("num_decoding_tokens", _i64),
                ("has_prefill", _int), ("prefill_block_tables", C.POINTER(Tensor)),
                ("max_prefill_sequence_length", _i64), ("query_start_locations", C.POINTER(Tensor)),
                ("sequence_start_locations", C.POINTER(Tensor)), ("max_sequence_length_k", _i64),
                ("has_decoding", _int), ("decoding_block_tables", C.POINTER(Tensor)),
                ("decoding_sequence_lengths", C.POINTER(Tensor))]


class FlashAttention(C.Structure):
    _fields_ = [("num_heads", _i64), ("num_kv_heads", _i64), ("head_dim", _i64), ("softmax_scale", _f32),
                ("alibi_slopes", C.POINTER(Tensor)), ("sliding_window", _i64), ("kv_cache_dtype", C.c_int32),
                ("device", C.c_int32)]


_TP = C.POINTER(Tensor)
_opt("atoma_flash_attn", [_TP, _TP, _TP, _f32, _int, _TP])
_opt("atoma_flash_attn_varlen", [_TP, _TP, _TP, _TP, _TP, _i64, _i64, _f32, _int, _TP])
_opt("atoma_flash_attn_varlen_with_block_table", [_TP, _TP, _TP, _TP, _TP, _TP, _i64, _i64, _f32, _i64, _i64, _TP, _TP])
_opt("atoma_flash_attn_kv_cache_full", [_TP, _TP, _TP, _TP, _f32, _TP, _TP, _int, _TP])
# the reference's other forms of the three ops (windows: negative = None)
_opt("atoma_flash_attn_windowed", [_TP, _TP, _TP, _f32, _i64, _i64, _TP])
_opt("atoma_flash_attn_alibi", [_TP, _TP, _TP, _TP, _f32, _int, _TP])
_opt("atoma_flash_attn_alibi_windowed", [_TP, _TP, _TP, _TP, _f32, _i64, _i64, _TP])
_opt("atoma_flash_attn_alibi_windowed_with_softcap", [_TP, _TP, _TP, _TP, _f32, _i64, _i64, _f32, _TP])
_opt("atoma_flash_attn_varlen_windowed", [_TP, _TP, _TP, _TP, _TP, _i64, _i64, _f32, _i64, _i64, _TP])
_opt("atoma_flash_attn_varlen_alibi", [_TP, _TP, _TP, _TP, _TP, _TP, _i64, _i64, _f32, _int, _TP])
_opt("atoma_flash_attn_varlen_alibi_windowed", [_TP, _TP, _TP, _TP, _TP, _TP, _i64, _i64, _f32, _i64, _i64, _TP])
_opt("atoma_flash_attn_varlen_full", [_TP, _TP, _TP, _TP, _TP, _TP, _i64, _i64, _f32, _i64, _i64, _TP, _TP, _f32, _TP])
_opt("atoma_flash_attn_kv_cache", [_TP, _TP, _TP, _f32, _int, _TP])
_opt("atoma_flash_attn_kv_cache_windowed", [_TP, _TP, _TP, _TP, _f32, _i64, _i64, _TP])
_opt("atoma_flash_attn_kv_cache_alibi", [_TP, _TP, _TP, _TP, _TP, _f32, _int, _TP])
_opt("atoma_flash_attn_kv_cache_alibi_windowed", [_TP, _TP, _TP, _TP, _f32, _i64, _i64, _TP])
_opt("atoma_reshape_and_cache_flash", [_TP, _TP, _TP, _TP, _TP])
_opt("atoma_copy_blocks", [C.POINTER(_TP), _i64, C.POINTER(_TP), _i64, _TP])
_opt("atoma_swap_blocks_tensor", [_TP, _TP, C.POINTER(C.c_uint32), _i64])
_opt("atoma_flash_attention_new", [C.POINTER(FlashAttention), _i64, _i64, _i64, _f32, _TP, _i64, C.c_int32, C.c_int32])
_opt("atoma_flash_attention_forward", [C.POINTER(FlashAttention), _TP, _TP, _TP, _TP, C.POINTER(AttnMetadata), _TP])


def ref(t):
    """byref() for an optional Tensor."""
    return None if t is None else C.byref(t)


def tensor_array(ts):
    arr = (_TP * len(ts))(*[C.pointer(t) for t in ts])
    return arr


# ---- host-side batch preparation (atoma_prepare_inputs) ----
class SeqDesc(C.Structure):
    _fields_ = [("is_prompt", C.c_int32), ("no_block_tables", C.c_int32), ("length", C.c_int64), ("num_computed_tokens", C.c_int64),
                ("token_chunk_size", C.c_int64), ("token_ids", C.POINTER(C.c_uint32)), ("block_table", C.POINTER(C.c_uint32)),
                ("block_table_len", C.c_int64)]


class BatchLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "num_sequences", "num_tokens", "num_slots", "num_prefills", "num_prefill_tokens", "num_decode_tokens", "max_query_len",
        "max_prefill_seq_len", "max_decode_seq_len", "max_block_table_len", "off_input_tokens", "off_input_positions", "off_slot_mapping",
        "off_seq_lens", "off_context_lens", "off_query_start_loc", "off_seq_start_loc", "off_block_tables", "total_bytes")]


_opt("atoma_prepare_inputs", [C.POINTER(SeqDesc), _i64, _i64, _i64, _int, _vp, _i64, _vp, _i64, C.POINTER(BatchLayout), _vp])


def make_seq_descs(seqs):
    """seqs: dicts as in oracle/batch_prep_oracle.py.  Returns (ctypes array, keep-alive list)."""
    arr = (SeqDesc * len(seqs))()
    keep = []
    for d, s in zip(arr, seqs):
        toks = np.ascontiguousarray(s["tokens"], np.uint32)
        keep.append(toks)
        d.is_prompt, d.no_block_tables = int(s["is_prompt"]), int(bool(s.get("no_block_tables")))
        d.length, d.num_computed_tokens, d.token_chunk_size = len(toks), int(s.get("num_computed", 0)), int(s["chunk"])
        d.token_ids = toks.ctypes.data_as(C.POINTER(C.c_uint32)) if len(toks) else None
        bt = s.get("block_table")
        if bt is not None:
            bt = np.ascontiguousarray(bt, np.uint32)
            keep.append(bt)
            d.block_table, d.block_table_len = bt.ctypes.data_as(C.POINTER(C.c_uint32)), len(bt)
    return arr, keep


def unpack_batch(buf, lay):
    """Views of the packed buffer (numpy uint8 array) as the tensors of ModelInput / FlashAttentionMetadata."""
    n = lay.num_sequences
    view = lambda off, count, dt: buf[off: off + count * np.dtype(dt).itemsize].view(dt)
    return dict(input_tokens=view(lay.off_input_tokens, lay.num_tokens, np.uint32), input_positions=view(lay.off_input_positions, lay.num_tokens, np.int64),
                slot_mapping=view(lay.off_slot_mapping, lay.num_slots, np.int64), seq_lens=view(lay.off_seq_lens, n, np.uint32),
                context_lens=view(lay.off_context_lens, n, np.uint32), query_start_loc=view(lay.off_query_start_loc, n + 1, np.uint32),
                seq_start_loc=view(lay.off_seq_start_loc, n + 1, np.uint32),
                block_tables=view(lay.off_block_tables, n * lay.max_block_table_len, np.uint32).reshape(n, lay.max_block_table_len),
                num_prefills=lay.num_prefills, num_prefill_tokens=lay.num_prefill_tokens, num_decode_tokens=lay.num_decode_tokens,
                max_query_len=lay.max_query_len, max_prefill_seq_len=lay.max_prefill_seq_len, max_decode_seq_len=lay.max_decode_seq_len)


def prepare_inputs_host(seqs, block_size, sliding_window=None, enable_chunked_prefill=False):
    """Pack on the host only (no device needed).  Returns (dict of arrays, layout) or raises RuntimeError(last_error())."""
    arr, keep = make_seq_descs(seqs)
    lay = BatchLayout()
    sw = int(sliding_window or 0)
    if lib.atoma_prepare_inputs(arr, len(seqs), block_size, sw, int(enable_chunked_prefill), None, 0, None, 0, C.byref(lay), None) != 0:
        raise RuntimeError(last_error())
    buf = np.zeros(lay.total_bytes, np.uint8)
    if lib.atoma_prepare_inputs(arr, len(seqs), block_size, sw, int(enable_chunked_prefill), buf.ctypes.data, buf.nbytes, None, 0,
                                C.byref(lay), None) != 0:
        raise RuntimeError(last_error())
    return unpack_batch(buf, lay), lay
