"""numpy restatement of the reference's stochastic token selection (TEST INFRASTRUCTURE, see oracle/__init__.py).

Reference: candle_transformers::generation::LogitsProcessor (candle-transformers 0.9.2-alpha.1, Cargo.lock; not in the tree),
configured at backends/vllm/src/llm_service.rs:348-372 and called at model_executor.rs:230-235.  PARITY UNPINNED: the crate's
source is absent and its random stream (rand StdRng) is not reproducible outside rand; what is restated is the published
algorithm -- softmax(logits / temperature) in f32, the top-k / top-p restrictions, and a weighted draw (WeightedIndex: one
uniform in [0, total), first index whose cumulative weight exceeds it) -- with the uniform supplied by the caller.  Our order of
the cumulative sum is (logit descending, index ascending) for the restricted variants (Candle's top-k order is unspecified).
Everything is evaluated in float64 here; `bracket` returns, per row, the interval of u for which a given token is the answer."""
import numpy as np


def kept_weights(logits_row, temperature, top_k=0, top_p=1.0):
    """Returns (token order, weights in that order) of the tokens that can be drawn."""
    x = np.asarray(logits_row, np.float64)
    x = np.where(np.isnan(x), -np.inf, x)
    w = np.exp((x - x.max()) / temperature)
    n = len(x)
    if (top_k <= 0 or top_k >= n) and top_p >= 1.0:
        return np.arange(n), w
    order = np.lexsort((np.arange(n), -x))                 # logit descending, index ascending
    if 0 < top_k < n:
        order = order[:top_k]
    ww = w[order]
    if top_p < 1.0:
        cum = np.cumsum(ww) / w.sum()
        reach = np.nonzero(cum >= top_p)[0]
        keep = reach[0] + 1 if len(reach) else len(ww)
        order, ww = order[:keep], ww[:keep]
    return order, ww


def sample(logits_row, u, temperature, top_k=0, top_p=1.0):
    order, w = kept_weights(logits_row, temperature, top_k, top_p)
    cum = np.cumsum(w)
    i = int(np.searchsorted(cum, u * cum[-1], side="right"))
    i = min(i, len(w) - 1)
    while w[i] == 0 and i > 0:
        i -= 1
    return int(order[i])


def bracket(logits_row, token, temperature, top_k=0, top_p=1.0):
    """(lo, hi): the token is the answer for u in [lo, hi) (empty when it cannot be drawn)."""
    order, w = kept_weights(logits_row, temperature, top_k, top_p)
    pos = np.nonzero(order == token)[0]
    if len(pos) == 0:
        return (1.0, 0.0)
    cum = np.cumsum(w)
    j = pos[0]
    return ((cum[j] - w[j]) / cum[-1], cum[j] / cum[-1])
