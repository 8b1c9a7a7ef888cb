"""First contact of the hand-scheduled prefill kernel (prefill_cfg = 4) with the hardware: small parity cases, then timings.
    python tools/probes/pfa_try.py check | bench
Run under `timeout`: a wrong barrier count hangs a workgroup."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
import numpy as np  # noqa: E402
import atoma_hip as ah  # noqa: E402
from oracle import attn_oracle as A  # noqa: E402
from oracle.halfs import BF16, F16, to_f32, from_f32  # noqa: E402
from util import rand_half  # noqa: E402
from test_attention_golden_gpu import gpu_varlen  # noqa: E402


def check(lens, h, hk, causal, dtype=BF16, exact_keys=512, lens_k=None, simple=0):
    rng = np.random.default_rng(1)
    d = 128
    lens = np.array(lens, np.int32)
    lk = lens if lens_k is None else np.array(lens_k, np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cuk = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
    q = rand_half(rng, (int(cu[-1]), h, d), dtype)
    k, v = rand_half(rng, (int(cuk[-1]), hk, d), dtype), rand_half(rng, (int(cuk[-1]), hk, d), dtype)
    ah.lib.atoma_set_option(b"prefill_cfg", 4)
    ah.lib.atoma_set_option(b"prefill_exact_keys", exact_keys)
    ah.lib.atoma_set_option(b"prefill_simple", simple)
    out, lse = gpu_varlen(ah, q, k, v, cu, cuk, d ** -0.5, causal, dtype)
    qf, kf, vf = to_f32(q, dtype), to_f32(k, dtype), to_f32(v, dtype)
    worst, nd, nan = 0.0, 0, 0
    for b in range(len(lens)):
        s0, s1, k0, k1 = int(cu[b]), int(cu[b + 1]), int(cuk[b]), int(cuk[b + 1])
        if s1 == s0:
            continue
        for name, pre in (("own", exact_keys == 0),):
            ref = from_f32(A.attend_prefill_online(qf[s0:s1], kf[k0:k1], vf[k0:k1], np.float32(d ** -0.5), causal, dtype, prescale=pre), dtype)
            got = to_f32(out[s0:s1], dtype)
            nan += int((~np.isfinite(got)).sum())
            err = np.abs(got - to_f32(ref, dtype))
            tol = 1e-3 + (2.0 ** -7 if dtype == BF16 else 2.0 ** -10) * np.abs(to_f32(ref, dtype))
            worst = max(worst, float(np.nanmax(err - tol)))
            nd += int((out[s0:s1] != ref).sum())
    print(f"check lens={list(lens)} lens_k={list(lk)} h={h}/{hk} causal={causal} dtype={dtype} exact_keys={exact_keys} simple={simple}: "
          f"worst (err - tol) {worst:.3e}  non-finite {nan}  differing patterns {nd} of {out.size}", flush=True)
    return worst <= 0 and nan == 0


def timeit(fn, iters=10):
    st = ah.Stream()
    for _ in range(2):
        fn(st)
    st.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn(st)
    st.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def bench():
    rng = np.random.default_rng(1)
    import tp_step as TS  # noqa
    h, hk, d = 32, 8, 128
    for S, nseq, causal in ((2048, 16, 1), (512, 16, 1), (4096, 4, 1), (2048, 16, 0)):
        T = S * nseq
        q, k, v = (TS.rand_dev(rng, T * n * d * 2) for n in (h, hk, hk))
        o = ah.DeviceBuffer(T * h * d * 2)
        cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
        flops = 4 * S * S * h * d * nseq / (2 if causal else 1)
        for cfg, ek, simple in ((0, 512, 0), (4, 512, 0), (4, 0, 0), (4, 0x7FFFFFFF, 0), (4, 512, 1)):
            ah.lib.atoma_set_option(b"prefill_cfg", cfg)
            ah.lib.atoma_set_option(b"prefill_exact_keys", ek)
            ah.lib.atoma_set_option(b"prefill_simple", simple)

            def run(st):
                ah.run_mha(q, k, v, o, b=nseq, h=h, h_k=hk, d=d, seqlen_q=S, seqlen_k=S, softmax_scale=d ** -0.5, is_bf16=1,
                           q_strides=(0, h * d, d), o_strides=(0, h * d, d), k_strides=(0, hk * d, d), v_strides=(0, hk * d, d),
                           is_causal=causal, cu_seqlens_q=cu, cu_seqlens_k=cu, stream=st.s)
            ms = min(timeit(run) for _ in range(3))
            print(f"bench S={S} x{nseq} causal={causal} cfg={cfg} exact_keys={ek} simple={simple}: {ms:.4f} ms  {flops / ms / 1e9:.1f} TF/s  "
                  f"frac {flops / ms / 1e9 / 2500:.3f}", flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    ah.set_device(0)
    if sys.argv[1] == "check":
        ok = True
        ok &= check([64], 1, 1, True)
        ok &= check([300], 2, 1, True)
        ok &= check([300], 2, 1, True, exact_keys=0)
        ok &= check([700], 2, 2, True, exact_keys=0)
        ok &= check([1, 31, 65, 129, 257, 600], 4, 2, True)
        ok &= check([1, 31, 65, 129, 257, 600], 4, 2, False, exact_keys=0)
        ok &= check([513], 2, 1, True, dtype=F16)
        ok &= check([100, 40], 2, 1, True, lens_k=[400, 10])
        ok &= check([1100], 4, 4, True, exact_keys=0, simple=1)
        print("ALL OK" if ok else "FAILED")
    else:
        bench()
