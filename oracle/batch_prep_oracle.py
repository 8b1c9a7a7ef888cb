"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ModelWorker::prepare_input_tensors
(/root/reference/backends/vllm/src/worker.rs:224-460) used to check atoma_prepare_inputs.  Parity unpinned: the
reference holds no test or golden vector for this function; the restatement follows its numbered steps line by line.

A sequence is a dict: is_prompt, tokens (all token ids of the sequence), num_computed (prompts), chunk
(token_chunk_size), block_table (list or None), no_block_tables (the group's block-table map is empty)."""
import numpy as np

PAD_SLOT_ID = -1   # worker.rs:13


def prepare_inputs(seqs, block_size, sliding_window=None, enable_chunked_prefill=False):
    input_tokens, input_positions, slot_mapping = [], [], []
    sequence_lengths, prefill_lens, decode_lens, context_lengths, query_lengths, block_tables = [], [], [], [], [], []
    num_prefills = num_prefill_tokens = num_decode_tokens = 0
    for s in seqs:
        is_prompt, toks = s["is_prompt"], list(s["tokens"])
        length = len(toks)
        context = s["num_computed"] if is_prompt else length - 1                      # 1.  worker.rs:248-257
        seq_len = min(length, context + s["chunk"])                                   # 2.  :260-262
        if is_prompt:                                                                 # 3.  :265-278
            tokens = toks[context:seq_len]
        else:
            if not toks:
                raise ValueError("Empty prompts should not be received in `ModelWorker`")
            tokens = [toks[-1]]
        query = seq_len - context if is_prompt else 1                                 # 4.  :281-285
        sliding_seq_len, sliding_context = seq_len, context                           # 5.  :293-306
        if sliding_window is not None and not is_prompt:
            sliding_seq_len = min(sliding_window, seq_len)
        if enable_chunked_prefill or not is_prompt:                                   # 6.  :309-332
            if s.get("block_table") is None:
                raise ValueError("Block table should be allocated for sequence on decoding phase")
            bt = list(s["block_table"])
            if sliding_window is not None:                                            # 7.  :319-325
                sw_blocks = (sliding_window + block_size - 1) // block_size
                bt = bt[max(0, len(bt) - sw_blocks):]
        else:
            bt = []
        block_tables.append(bt)                                                       # 8.  :335-341
        sequence_lengths.append(sliding_seq_len)
        context_lengths.append(sliding_context)
        query_lengths.append(query)
        input_tokens.extend(tokens)
        input_positions.extend(range(context, seq_len))
        if is_prompt:                                                                 # 9.  :345-362
            num_prefills += 1
            num_prefill_tokens += len(tokens)
            prefill_lens.append(seq_len)
        else:
            num_decode_tokens += query
            decode_lens.append(sliding_seq_len)
        if s.get("no_block_tables"):                                                  #     :364-371
            slot_mapping.extend([PAD_SLOT_ID] * seq_len)
            continue
        table = s["block_table"]                                                      # 10. :374-399
        start_index = max(0, query - sliding_window) if sliding_window is not None else 0
        for i in range(context, seq_len):
            slot_mapping.append(PAD_SLOT_ID if i < start_index else table[i // block_size] * block_size + i % block_size)
    max_bt = max(len(b) for b in block_tables)                                        # 11. :403-441
    bt_tensor = np.zeros((len(block_tables), max_bt), np.uint32)
    for i, b in enumerate(block_tables):
        bt_tensor[i, :len(b)] = b
    cum = lambda v: np.concatenate([[0], np.cumsum(np.asarray(v, np.float32)).astype(np.uint32)]).astype(np.uint32)
    return dict(
        input_tokens=np.asarray(input_tokens, np.uint32), input_positions=np.asarray(input_positions, np.int64),
        slot_mapping=np.asarray(slot_mapping, np.int64), seq_lens=np.asarray(sequence_lengths, np.uint32),
        context_lens=np.asarray(context_lengths, np.uint32), query_start_loc=cum(query_lengths), seq_start_loc=cum(sequence_lengths),
        block_tables=bt_tensor, num_prefills=num_prefills, num_prefill_tokens=num_prefill_tokens, num_decode_tokens=num_decode_tokens,
        max_query_len=max(query_lengths, default=0), max_prefill_seq_len=max(prefill_lens, default=0),
        max_decode_seq_len=max(decode_lens, default=0))
