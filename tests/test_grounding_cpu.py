"""Row f3 (second half): scheduling logic of RecursiveGroundingBatcher on CPU with a stand-in encoder -- the
micro-batched walk gives every example exactly the bounds / answers / clips of the one-example-at-a-time loop
(oracle/grounding_oracle.py, restating tasks/test_recursive_grounding.py:350-466)."""
import zlib

import numpy as np
import torch

from hawkeye_amd import grounding as G
from oracle import grounding_oracle as GO


def _fake_encode(frames, instruction):
    # a deterministic function of the clip's bytes, row by row (stands in for the HIP path on CPU)
    f = frames.to(torch.float32)
    return (f.mean(dim=(1, 2, 3, 4))[:, None, None] + f[:, :, :4, :4, 0].mean(dim=(1, 2, 3))[:, None, None]
            ).expand(-1, 4, 8).contiguous(), False


def _answer(example, turn, out, bound):
    # depends on the encoder output of THIS clip and on the turn: a wrong clip or a wrong bound changes the walk
    h = zlib.crc32(np.asarray(out.detach().float().cpu().numpy()).tobytes()) + 31 * turn + example
    return G.CHOICES[h % 4]


def test_frame_indices_and_narrow_match_the_loop_restatement():
    rng = np.random.default_rng(0)
    for _ in range(300):
        fps = float(rng.choice([3.0, 23.976, 25.0, 29.97, 30.0]))
        nfr = int(rng.integers(13, 4000))
        nseg = int(rng.choice([8, 12, 16]))
        if rng.random() < 0.2:
            bound = None
        else:
            a = float(rng.uniform(0, nfr / fps * 0.9))
            bound = (a, float(rng.uniform(a, nfr / fps)))
        assert np.array_equal(G.frame_indices(bound, fps, nfr - 1, nseg), GO.get_index(bound, fps, nfr - 1, nseg))
    for ans in G.CHOICES:
        assert list(G.narrow((3.0, 11.0), ans)) == list(GO.next_bound([3.0, 11.0], ans))


def test_micro_batched_walk_equals_one_example_at_a_time():
    src = G.SyntheticFrameSource(size=16, seed=3)
    examples = list(range(23))
    ins = (torch.tensor([[101, 7, 102]]), torch.tensor([[1, 1, 1]]))
    seen_batches = []

    def enc(frames, instruction):
        seen_batches.append((frames.shape[0], tuple(instruction[0].shape)))
        return _fake_encode(frames, instruction)

    b = G.RecursiveGroundingBatcher(enc, src, _answer, ins, max_turns=4, num_segments=6, micro_batch=8, size=16,
                                    device="cpu", keep_outputs=True)
    got = b.run(examples)
    assert max(n for n, _ in seen_batches) == 8 and all(s == (n, 3) for n, s in seen_batches)
    assert b.stats["clips"] == sum(len(v["answers"]) for v in got.values())
    assert b.stats["encode_calls"] < b.stats["clips"] / 3          # really batched
    n_multi = 0
    for ex in examples:
        bound, answers, duration, outs = GO.inference(
            ex, src, lambda fr: _fake_encode(torch.from_numpy(fr), None)[0], _answer, 4, 6)
        assert got[ex]["answers"] == answers and tuple(got[ex]["bound"]) == bound and got[ex]["duration"] == duration
        assert all(torch.equal(a, o) for a, o in zip(got[ex]["outputs"], outs))
        assert len(answers) <= 4 and (answers[-1] == "throughout" or len(answers) == 4)
        n_multi += len(answers) > 1
    assert n_multi > 5
    # example partition over ranks = --start-idx/--end-idx blocks (tasks/test_recursive_grounding.py:481-482)
    parts = [G.RecursiveGroundingBatcher(_fake_encode, src, _answer, ins, max_turns=4, num_segments=6, micro_batch=8,
                                         size=16, device="cpu").run(examples, rank=r, world=3) for r in range(3)]
    merged = {}
    for p in parts:
        assert not set(p) & set(merged)
        merged.update(p)
    assert sorted(merged) == examples and all(merged[e]["answers"] == got[e]["answers"] for e in examples)
