#!/usr/bin/env python3
"""Where do the 8-rank and the unsharded prefill chunk part ways at the full configs[3] shapes?  One layer, T tokens; the shards are cut
on the device from the full model; every intermediate of the unsharded step against what the ranks hold.
usage: python tools/probes/tp_prefill_debug.py [T] [layers]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv.insert(1, "--virtual-ranks")          # (tp_step.py sets GPU_MAX_HW_QUEUES when it sees this)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tp_step as TS  # noqa: E402
sys.argv.pop(1)
ah, DS, tp = TS.ah, TS.DS, TS.tp
from halfs import BF16, to_f32  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = DS.Config(layers, 8192, 64, 8, 128, 28672, 128256)
W = 8
ah.set_device(0)
full_w = TS.random_shard_weights(np.random.default_rng(11), cfg)
ranks = []
for r in range(W):
    rk = TS.Rank(cfg, r, W, 64, 4096, 0, weights=TS.shard_device_weights(full_w, cfg, r, W), prefill=T)
    ranks.append(rk)
un = TS.Rank(cfg, 0, 1, 64, 4096, 0, weights=full_w, prefill=T)
xs = []
for r in range(W):
    h = C.c_void_p()
    assert ah.lib.atoma_xgmi_create(C.byref(h), r, W, 0, T * 8192 * 2) == 0, ah.last_error()
    xs.append(h)
blobs = (C.c_uint8 * (128 * W))()
for r in range(W):
    one = (C.c_uint8 * 128)()
    assert ah.lib.atoma_xgmi_handle(xs[r], one) == 0
    C.memmove(C.addressof(blobs) + 128 * r, one, 128)
for r in range(W):
    assert ah.lib.atoma_xgmi_connect(xs[r], blobs) == 0, ah.last_error()
for rk in ranks:                       # vendor-GEMM plans with the exchange off
    rk.engine = lambda p, c: None
    rk.step.run()
    rk.stream.synchronize()
for r, rk in enumerate(ranks):
    rk.engine = (lambda p, c, r=r, rk=rk: ah.lib.atoma_xgmi_allreduce_sum(xs[r], p, p, c, BF16, rk.stream.s))
for rk in ranks:
    rk.step.run()
for rk in ranks:
    rk.stream.synchronize()
print("xgmi status", [ah.lib.atoma_xgmi_status(x) for x in xs])
un.step.run()
un.stream.synchronize()


def f32(buf, shape):
    return to_f32(buf.numpy(np.uint16, shape), BF16)


def cmp(name, a, b):
    d = np.abs(a - b)
    print(f"{name:34s} max|diff| {d.max():9.4f}   rms(ref) {np.sqrt((b * b).mean()):8.4f}   frac > 0.1: {(d > 0.1).mean():.5f}   worst row {np.unravel_index(d.argmax(), d.shape)}")


H, c, sc = 8192, cfg, ranks[0].c
uq = f32(un.step.qkv, (T, c.qkv))
ua = f32(un.step.att, (T, c.h * c.d))
for r in (0, 3, 7):
    rq = f32(ranks[r].step.qkv, (T, sc.qkv))
    cmp(f"rank {r} q (post-RoPE)", rq[:, :sc.h * c.d], uq[:, r * sc.h * c.d:(r + 1) * sc.h * c.d])
    cmp(f"rank {r} k (post-RoPE)", rq[:, sc.h * c.d:(sc.h + 1) * c.d], uq[:, (c.h + r) * c.d:(c.h + r + 1) * c.d])
    cmp(f"rank {r} v", rq[:, (sc.h + 1) * c.d:], uq[:, (c.h + c.hk + r) * c.d:(c.h + c.hk + r + 1) * c.d])
    cmp(f"rank {r} attention out", f32(ranks[r].step.att, (T, sc.h * c.d)), ua[:, r * sc.h * c.d:(r + 1) * sc.h * c.d])
if layers == 1:
    cmp("down-proj output (all-reduced)", f32(ranks[0].step.o, (T, H)), f32(un.step.o, (T, H)))
    cmp("gate/up act", np.concatenate([f32(rk.step.act, (T, sc.inter)) for rk in ranks], 1), f32(un.step.act, (T, c.inter)))
for nm in ("x", "x1", "x2", "xn"):
    cmp(f"residual buffer {nm}", f32(getattr(ranks[0].step, nm), (T, H)), f32(getattr(un.step, nm), (T, H)))
cmp("final norm input row (xf)", f32(ranks[0].step.xf, (1, H)), f32(un.step.xf, (1, H)))
cmp("logits (last token)", f32(ranks[0].step.logits, (1, c.vocab)), f32(un.step.logits, (1, c.vocab)))
for x in xs:
    ah.lib.atoma_xgmi_destroy(x)
