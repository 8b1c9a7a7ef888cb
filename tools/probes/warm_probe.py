"""Does the measured time of a dense kernel depend on how long the GPU has been busy before the timed region?  (clock ramp / power state)
    python tools/probes/warm_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    sys.path.insert(0, p)
import numpy as np
import atoma_hip as ah
import tp_step as TS

ah.set_device(0)
rng = np.random.default_rng(1)
h, hk, d, S, nseq = 32, 8, 128, 2048, 16
T = S * nseq
q, k, v = (TS.rand_dev(rng, T * n * d * 2) for n in (h, hk, hk))
o = ah.DeviceBuffer(T * h * d * 2)
cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
st = ah.Stream()


def run():
    ah.run_mha(q, k, v, o, b=nseq, h=h, h_k=hk, d=d, seqlen_q=S, seqlen_k=S, softmax_scale=d ** -0.5, is_bf16=1, q_strides=(0, h * d, d),
               o_strides=(0, h * d, d), k_strides=(0, hk * d, d), v_strides=(0, hk * d, d), is_causal=1, cu_seqlens_q=cu, cu_seqlens_k=cu, stream=st.s)


for cfg in (4, 0, 4, 0):
    ah.lib.atoma_set_option(b"prefill_cfg", cfg)
    for warm, idle in ((2, 0.5), (10, 0.5), (50, 0.5), (200, 0.5), (2, 0.0)):
        time.sleep(idle)
        for _ in range(warm):
            run()
        a, b = ah.Event(), ah.Event()
        a.record(st.s)
        for _ in range(10):
            run()
        b.record(st.s)
        b.synchronize()
        ms = a.elapsed_ms(b) / 10
        print(f"cfg {cfg} idle {idle}s warm {warm:3d}: {ms:.4f} ms  {4 * S * S * h * d / 2 * nseq / ms / 1e9:.0f} TF/s", flush=True)
