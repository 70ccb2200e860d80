"""ORACLE -- test infrastructure, not product code (see oracle/encode_img_oracle.py for the rules).

One-example-at-a-time restatement of the recursive grounding loop, tasks/test_recursive_grounding.py:350-362
(get_index), :409-423 (inference) and :425-466 (single_turn: the bound arithmetic), as plain Python loops.
PARITY UNPINNED for the index / bound arithmetic: the script it restates imports decord, torchvision, cv2, peft,
matplotlib and IPython at module level, none of which exist in the build container, so it cannot be executed to
produce fixtures; the functions are ~20 lines of scalar arithmetic and are restated line for line.  The tensors
that flow through (encode_img outputs) are pinned elsewhere (tests/golden/full_T12_B2.npz etc.)."""
import numpy as np


def get_index(bound, fps, max_frame, num_segments, first_idx=0):
    """:350-362"""
    if bound:
        start, end = bound[0], bound[1]
    else:
        start, end = -100000, 100000
    start_idx = max(first_idx, round(start * fps))
    end_idx = min(round(end * fps), max_frame)
    seg_size = float(end_idx - start_idx) / num_segments
    return np.array([int(start_idx + (seg_size / 2) + np.round(seg_size * idx)) for idx in range(num_segments)])


def next_bound(bound, pred_answer):
    """:456-464"""
    mid_sec, quarter_length = (bound[1] + bound[0]) / 2, (bound[1] - bound[0]) / 4
    if pred_answer == 'throughout':
        pred_bound = bound
    elif pred_answer == 'beginning':
        pred_bound = [bound[0], mid_sec]
    elif pred_answer == 'end':
        pred_bound = [mid_sec, bound[1]]
    elif pred_answer == 'middle':
        pred_bound = [mid_sec - quarter_length, mid_sec + quarter_length]
    return pred_bound


def inference(example, source, encode_one, answer_fn, max_turns, num_segments):
    """:409-423 with single_turn (:425-466) inlined: one clip per call, B = 1.
    encode_one(frames uint8 [1,T,H,W,3]) -> [1, nq, C]."""
    fps, nfr = source.info(example)
    duration = nfr / fps
    bound, answers, outs = None, [], []
    for turn in range(max_turns):
        idx = get_index(bound, fps, nfr - 1, num_segments)
        buf = np.empty((num_segments, source.size, source.size, 3), dtype=np.uint8)
        source.frames(example, idx, buf)
        out = encode_one(buf[None])
        if bound is None:
            bound = [0, duration]
        ans = answer_fn(example, turn, out[0], tuple(bound))
        answers.append(ans)
        outs.append(out[0])
        bound = next_bound(bound, ans)
        if ans == 'throughout':
            break
    return tuple(bound), answers, duration, outs
