#!/usr/bin/env python3
"""Launch-to-launch spread of the hand-scheduled prefill kernel (493-699 us, sigma 43 us on 525 us): N back-to-back launches of the
16 x 2048 causal workload, every one bracketed by its own pair of events on the launch stream (no host sync in between), after the usual
60 ms warm-up.  Prints the distribution and the series, so that a drift (clocks), a period (power management) or outliers show."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "atoma-infer_amd", "bindings")):
    sys.path.insert(0, p)
import atoma_hip as ah  # noqa: E402
import tp_step as TS  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S, nseq, h, hk, d = 2048, 16, 32, 8, 128
ah.set_device(0)
rng = np.random.default_rng(1)
T = S * nseq
q, k, v = (TS.rand_dev(rng, T * n * d * 2) for n in (h, hk, hk))
o = ah.DeviceBuffer(T * h * d * 2)
cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
st = ah.Stream()


def run():
    ah.run_mha(q, k, v, o, b=nseq, h=h, h_k=hk, d=d, seqlen_q=S, seqlen_k=S, softmax_scale=d ** -0.5, is_bf16=1, q_strides=(0, h * d, d), o_strides=(0, h * d, d),
               k_strides=(0, hk * d, d), v_strides=(0, hk * d, d), is_causal=1, cu_seqlens_q=cu, cu_seqlens_k=cu, stream=st.s)


import time
t0 = time.perf_counter()
while (time.perf_counter() - t0) < 0.1:
    run()
    st.synchronize()
ev = [ah.Event() for _ in range(N + 1)]
ev[0].record(st.s)
for i in range(N):
    run()
    ev[i + 1].record(st.s)
ev[-1].synchronize()
ms = np.array([ev[i].elapsed_ms(ev[i + 1]) for i in range(N)])
flops = 4 * S * S * h * d / 2 * nseq
print(f"launches {N}: min {ms.min() * 1e3:.1f} us  median {np.median(ms) * 1e3:.1f}  mean {ms.mean() * 1e3:.1f}  max {ms.max() * 1e3:.1f}  sigma {ms.std() * 1e3:.1f}  "
      f"(plan + persistent kernel per launch; median = {flops / np.median(ms) / 1e9:.0f} TF/s)")
print("deciles (us):", [round(float(x) * 1e3, 1) for x in np.percentile(ms, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])])
print("series, means of 10 consecutive launches (us):", [round(float(x) * 1e3, 1) for x in ms[: N // 10 * 10].reshape(-1, 10).mean(1)])
print("first 30 (us):", [round(float(x) * 1e3, 1) for x in ms[:30]])
