"""Row f3 (second half) on a real MI355X: RecursiveGroundingBatcher over model.encode_frames -- uint8 frames through
pinned double buffers and a copy stream, micro-batches across examples -- against the one-example-at-a-time loop
(oracle/grounding_oracle.py restating tasks/test_recursive_grounding.py:409-466) driving the same model at B = 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from hawkeye_amd import grounding as G  # noqa: E402
from hawkeye_amd import posembed, weights as W  # noqa: E402
from hawkeye_amd.config import SMALL  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402
from oracle import grounding_oracle as GO  # noqa: E402

DEV = "cuda:0"


def _answer(example, turn, out, bound):
    # host-visible decision that does not hinge on the last bits of the output (B = 1 and B = 8 take different GEMM
    # kernels): the walk is decided by (example, turn); the outputs themselves are compared below
    return G.CHOICES[(example * 7 + turn * 3 + (example >> 2)) % 4]


def test_micro_batched_grounding_equals_one_clip_per_call():
    cfg, seed, T = SMALL, 31, 3
    m = HawkEyeEncoder(cfg, out_dtype=torch.float32)
    m.load_state_dict(W.make_state_dict(cfg, seed), strict=True)
    m = m.to(DEV).eval()
    m.vision_encoder.encoder.pos_embed = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim,
                                                                  cfg.ckpt_num_frame)
    ids, mask = W.make_instruction_ids(1, 6, cfg.qf_vocab, seed)
    ins = (ids.to(DEV), mask.to(DEV))
    src = G.SyntheticFrameSource(size=cfg.img_size, seed=5)
    examples = list(range(19))
    with torch.no_grad():
        b = G.RecursiveGroundingBatcher(m.encode_frames, src, _answer, ins, max_turns=4, num_segments=T, micro_batch=8,
                                        size=cfg.img_size, device=DEV, keep_outputs=True)
        got = b.run(examples)
        assert b.stats["encode_calls"] < b.stats["clips"] / 3 and b.stats["waves"] == 4
        worst = 1.0
        for ex in examples:
            bound, answers, duration, outs = GO.inference(
                ex, src, lambda fr: m.encode_frames(torch.from_numpy(fr).to(DEV), ins)[0].float().cpu(), _answer, 4, T)
            assert got[ex]["answers"] == answers and tuple(got[ex]["bound"]) == bound
            assert len(got[ex]["outputs"]) == len(outs)
            for a, o in zip(got[ex]["outputs"], outs):
                c = float((a.double().flatten() @ o.double().flatten()) / (a.double().norm() * o.double().norm()))
                worst = min(worst, c)
        print(f"[grounding] {b.stats['clips']} clips in {b.stats['encode_calls']} calls; worst cosine vs B=1 path {worst:.7f}")
        assert worst > 0.99999
