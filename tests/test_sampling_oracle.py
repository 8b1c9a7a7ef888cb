"""oracle/sampling_oracle.py on the CPU: the weighted draw, the top-k / top-p restrictions and the u-interval helper."""
import numpy as np

from oracle import sampling_oracle as SO


def test_weighted_draw_and_restrictions():
    x = np.log(np.array([0.1, 0.4, 0.2, 0.3]))
    assert [SO.sample(x, u, 1.0) for u in (0.0, 0.09, 0.11, 0.49, 0.51, 0.69, 0.71, 0.999)] == [0, 0, 1, 1, 2, 2, 3, 3]
    # top-k = 2 keeps tokens 1 (0.4) and 3 (0.3), most probable first
    assert [SO.sample(x, u, 1.0, top_k=2) for u in (0.0, 0.56, 0.58, 0.99)] == [1, 1, 3, 3]
    # top-p = 0.65: 0.4, then 0.3 reaches 0.7 >= 0.65 and is kept, the rest is dropped
    order, w = SO.kept_weights(x, 1.0, top_p=0.65)
    assert order.tolist() == [1, 3]
    assert SO.bracket(x, 3, 1.0, top_p=0.65) == (0.4 / 0.7, 1.0) or abs(SO.bracket(x, 3, 1.0, top_p=0.65)[0] - 0.4 / 0.7) < 1e-12
    assert SO.bracket(x, 0, 1.0, top_k=2) == (1.0, 0.0)
    # temperature sharpens
    o, w2 = SO.kept_weights(x, 0.5)
    assert np.allclose(w2 / w2.sum(), np.array([0.01, 0.16, 0.04, 0.09]) / 0.30)
    # NaN and -inf are never drawn
    y = np.array([np.nan, 0.0, -np.inf, 0.0])
    assert set(SO.sample(y, u, 1.0) for u in np.linspace(0, 0.999, 50)) == {1, 3}
