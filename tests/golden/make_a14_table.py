"""Pin table for SURVEY.md section 8a row A14 (the host decode step of GenerationSession.decode): the step-dependent host
tensors the reference feeds the engine, for B = 2, input_lengths = [3, 5], max_new_tokens = 4 (max_input_length = 5,
max_seq_length = 9), gpt-attention plugin on, padded inputs, beam width 1.

Pure integer logic restated from T/tensorrt_llm/runtime/generation.py (no import: the module needs tensorrt):
  :808-811  sequence_lengths = full(max_input_length)                      (dynamic decoder state, before step 0)
  :812-821  masked_tokens[b, t] = 1 for input_lengths[b] <= t < max_input_length
  :735-750  context:     position_ids = arange(max_input_length) per row ; last_token_ids = input_lengths
  :576-579  context:     sequence_length = max_input_length + step (step = 0) ; past_key_value_length = [0, 1]
  :752-767  generation:  position_ids = input_lengths + step ; last_token_ids = ones
  :686-689  generation:  sequence_length = max_input_length + step ; past_key_value_length = [max_input_length + step, 0]
  :852-946  loop: iteration `step` RUNS with the tensors prepared at the tail of iteration step - 1 (`step` there = step - 1);
            iteration 0 runs the context tensors.  The last iteration prepares nothing (:925).
Writes tests/golden/a14_host_step_table.json.  Run: python tests/golden/make_a14_table.py"""
import json
import os

B, LENS, MAX_NEW = 2, [3, 5], 4
MAX_IN = max(LENS)
SMAX = MAX_IN + MAX_NEW

masked = [[1 if LENS[b] <= t < MAX_IN else 0 for t in range(SMAX)] for b in range(B)]
runs = []
# run 0: the context phase
runs.append(dict(run=0, phase='context', sequence_length=[MAX_IN + 0] * B, past_key_value_length=[0, 1],
                 position_ids=[list(range(MAX_IN)) for _ in range(B)], last_token_ids=list(LENS)))
# run k >= 1: generation, prepared with step = k - 1
for k in range(1, MAX_NEW):
    step = k - 1
    runs.append(dict(run=k, phase='generation', sequence_length=[MAX_IN + step] * B, past_key_value_length=[MAX_IN + step, 0],
                     position_ids=[[LENS[b] + step] for b in range(B)], last_token_ids=[1] * B))
table = dict(batch_size=B, input_lengths=LENS, max_input_length=MAX_IN, max_new_tokens=MAX_NEW, max_seq_length=SMAX,
             masked_tokens=masked, runs=runs,
             # slot the token consumed by run k is written to, and the attention span it sees (MM/...Template.h:1323-1330,
             # :1425-1426: slot = past length, rotary position = past length - (max_input_length - input_length))
             kv_slot_written=[None] + [MAX_IN + k - 1 for k in range(1, MAX_NEW)])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'a14_host_step_table.json')
json.dump(table, open(out, 'w'), indent=1)
print('wrote', out)
