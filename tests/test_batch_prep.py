"""atoma_prepare_inputs (host logic, runs without a GPU) against the restatement of ModelWorker::prepare_input_tensors
(backends/vllm/src/worker.rs:224-460): integer work, bit-exact."""
import numpy as np
import pytest

import atoma_hip as ah
from oracle import batch_prep_oracle as BO

KEYS_ARRAY = ("input_tokens", "input_positions", "slot_mapping", "seq_lens", "context_lens", "query_start_loc", "seq_start_loc", "block_tables")
KEYS_SCALAR = ("num_prefills", "num_prefill_tokens", "num_decode_tokens", "max_query_len", "max_prefill_seq_len", "max_decode_seq_len")


def random_batch(rng, n, block_size, chunked, p_prompt=0.3, max_len=300):
    seqs = []
    free = list(rng.permutation(n * (max_len // block_size + 2)))
    for _ in range(n):
        length = int(rng.integers(1, max_len))
        toks = rng.integers(0, 128256, length)
        pages = [int(free.pop()) for _ in range((length + block_size - 1) // block_size)]
        if rng.random() < p_prompt:
            computed = int(rng.integers(0, length)) if chunked else 0
            chunk = int(rng.integers(1, length - computed + 1)) if chunked else length
            seqs.append(dict(is_prompt=True, tokens=toks, num_computed=computed, chunk=chunk, block_table=pages))
        else:
            seqs.append(dict(is_prompt=False, tokens=toks, num_computed=length - 1, chunk=1, block_table=pages))
    return seqs


def same(got, ref):
    for k in KEYS_ARRAY:
        assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        assert np.array_equal(got[k], ref[k]), k
    for k in KEYS_SCALAR:
        assert int(got[k]) == int(ref[k]), k


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("chunked,sliding", [(False, None), (True, None), (False, 64), (True, 40)])
def test_prepare_inputs_matches_the_reference_restatement(seed, chunked, sliding):
    rng = np.random.default_rng(seed)
    block = int(rng.choice([16, 32]))
    seqs = random_batch(rng, int(rng.integers(1, 40)), block, chunked)
    got, lay = ah.prepare_inputs_host(seqs, block, sliding, chunked)
    same(got, BO.prepare_inputs(seqs, block, sliding, chunked))
    offs = [getattr(lay, n) for n, _ in ah.BatchLayout._fields_ if n.startswith("off_")]
    assert all(o % 256 == 0 for o in offs) and offs == sorted(offs) and lay.total_bytes % 256 == 0


def test_prepare_inputs_decode_batch_of_256_and_profiling_run():
    """The C3 shape: 256 decode sequences; and a memory-profiling run (no block tables: slots padded with -1)."""
    rng = np.random.default_rng(11)
    seqs = random_batch(rng, 256, 16, False, p_prompt=0.0, max_len=4096)
    got, lay = ah.prepare_inputs_host(seqs, 16)
    same(got, BO.prepare_inputs(seqs, 16))
    assert lay.num_tokens == 256 and lay.num_decode_tokens == 256 and lay.num_prefills == 0
    assert got["block_tables"].shape == (256, max(len(s["block_table"]) for s in seqs))
    prof = [dict(is_prompt=True, tokens=rng.integers(0, 100, 37), num_computed=0, chunk=37, block_table=None, no_block_tables=True)
            for _ in range(3)]
    got, _ = ah.prepare_inputs_host(prof, 16)
    same(got, BO.prepare_inputs(prof, 16))
    assert (got["slot_mapping"] == -1).all() and got["block_tables"].shape == (3, 0)


def test_prepare_inputs_errors():
    toks = np.arange(5)
    with pytest.raises(RuntimeError, match="Empty prompts should not be received in `ModelWorker`"):
        ah.prepare_inputs_host([dict(is_prompt=False, tokens=[], chunk=1, block_table=[0])], 16)
    with pytest.raises(RuntimeError, match="Block table should be allocated for sequence on decoding phase"):
        ah.prepare_inputs_host([dict(is_prompt=False, tokens=toks, chunk=1, block_table=None)], 16)
    with pytest.raises(RuntimeError, match="block table too short"):
        ah.prepare_inputs_host([dict(is_prompt=False, tokens=np.arange(40), chunk=1, block_table=[3, 4])], 16)
    with pytest.raises(RuntimeError, match="the batch is empty"):
        ah.prepare_inputs_host([], 16)
