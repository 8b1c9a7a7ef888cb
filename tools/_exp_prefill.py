import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_kernels as bk, numpy as np
ah = bk.ah
ah.set_device(0)
rng = np.random.default_rng(1)
h, hk, d = 32, 8, 128
S, nseq = 2048, 16
T = S * nseq
q, k, v = bk.rand_dev(rng, T * h * d * 2), bk.rand_dev(rng, T * hk * d * 2), bk.rand_dev(rng, T * hk * d * 2)
o = ah.DeviceBuffer(T * h * d * 2)
cu = ah.DeviceBuffer.from_numpy((np.arange(nseq + 1) * S).astype(np.int32))
def run():
    ah.run_mha(q, k, v, o, b=nseq, h=h, h_k=hk, d=d, seqlen_q=S, seqlen_k=S, softmax_scale=d ** -0.5, is_bf16=1,
               q_strides=(0, h * d, d), o_strides=(0, h * d, d), k_strides=(0, hk * d, d), v_strides=(0, hk * d, d),
               is_causal=1, cu_seqlens_q=cu, cu_seqlens_k=cu)
print(bk.timeit(run, iters=10))
