"""`tensorrt_llm.runtime.Session` (T/tensorrt_llm/runtime/session.py:37-190): deserialise an engine, ask for the output shapes,
run the context phase.  The engine is the traced tiny LLaMA of the HF golden fixture (tests/test_frontend.build_tiny_engine);
the logits are checked against HF's, and bad inputs make run() answer False / infer_shapes answer None instead of
raising an assertion from inside the native session."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


@pytest.fixture(scope='module')
def session():
    from test_frontend import build_tiny_engine
    from tensorrt_llm.runtime import Session
    engine, _, t = build_tiny_engine()
    s = Session.from_serialized_engine(engine)
    yield s, t


def test_infer_shapes_reports_the_engines_vocabulary(session):
    from tensorrt_llm.runtime.session import TensorInfo
    s, _ = session
    out = s.infer_shapes([TensorInfo('input_ids', np.int32, (2, 8)), TensorInfo('input_lengths', np.int32, (2, ))])
    assert [(o.name, tuple(o.shape)) for o in out] == [('logits', (2, 128))]  # vocab_size of the fixture, not -1
    assert s.infer_shapes([TensorInfo('position_idz', np.int32, (2, 8))]) is None          # not an input of the engine
    assert s.infer_shapes([TensorInfo('input_ids', np.float32, (2, 8))]) is None           # wrong dtype
    assert s.infer_shapes([TensorInfo('input_ids', np.int32, (16, ))]) is None             # not [batch, len]
    assert s.infer_shapes([TensorInfo('input_lengths', np.int32, (2, ))]) is None          # no input_ids at all


def test_run_matches_hf_and_rejects_bad_inputs(session):
    s, t = session
    ids = t['ids'].astype(np.int32)
    B, S = ids.shape
    lens = t['input_lengths'].astype(np.int32)
    out = np.zeros((B, 128), np.float32)
    assert s.run({'input_ids': ids, 'input_lengths': lens}, {'logits': out}) is True
    np.testing.assert_allclose(out, t['logits_ctx'], atol=1e-1)  # the reference's fp16 tolerance (T/tests/model/test_llama.py:288)
    good = out.copy()
    # bad inputs: False, nothing raised, the output untouched
    out[...] = 7.0
    assert s.run({'input_ids': ids}, {'logits': out}) is False                                           # a tensor missing
    assert s.run({'input_ids': ids, 'input_lengths': lens[:1]}, {'logits': out}) is False                # lengths of another batch
    assert s.run({'input_ids': ids.ravel(), 'input_lengths': lens}, {'logits': out}) is False            # not 2-D
    assert s.run({'input_ids': ids, 'input_lengths': lens + S}, {'logits': out}) is False                # longer than the buffer
    assert s.run({'input_ids': ids, 'input_lengths': lens}, {'logits': np.zeros((B, 64), np.float32)}) is False  # wrong vocab
    assert s.run({'input_ids': ids.astype(np.float32), 'input_lengths': lens}, {'logits': out}) is False  # wrong dtype
    assert np.all(out == 7.0)
    # and the session still works afterwards
    assert s.run({'input_ids': ids, 'input_lengths': lens}, {'logits': out}) is True
    np.testing.assert_array_equal(out, good)
