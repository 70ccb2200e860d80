"""Row f3, second half: the recursive-grounding access pattern (BASELINE config 3) fed to the HIP path in micro-batches.

The reference (tasks/test_recursive_grounding.py:409-466) walks ONE example at a time: up to `max_turns` times it
decodes `num_segments` frames of the current sub-segment, transforms them on the CPU, copies a [1,T,3,224,224] fp32
clip to the GPU, runs encode_img at B=1, lets the LLM pick beginning / middle / end / throughout and narrows the
segment.  Turn k+1 of an example depends on turn k's answer, but examples are independent, so

  * `RecursiveGroundingBatcher` advances many examples together: one "wave" = the next turn of every example that is
    still active, cut into micro-batches of up to `micro_batch` clips -> ONE encode_frames call each (B = 8..32
    instead of 1: 500.6 clips/s at micro-batches of 32 against 191.3 one clip per call, profiles/r03_cfg3_grounding.json);
  * frames travel as uint8 [B,T,H,W,3] (4x fewer bytes than the fp32 clip; /255, mean, std happen inside the patch
    gather, hk_patchify_u8) through two PINNED host buffers and two device buffers: while the GPU encodes micro-batch
    k, the host decodes micro-batch k+1 into the other pinned buffer and a side stream copies it over;
  * examples are block-partitioned over ranks exactly as --start-idx/--end-idx would (hawkeye_amd.dist.shard_range).

Video decoding itself (decord) and the bicubic resize stay on the host and out of scope: a `FrameSource` hands over
decoded, resized, cropped uint8 frames.  The LLM between the turns is a callable (`answer_fn`): row f1 covers its
prefill, generation is out of scope.
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

CHOICES = ("beginning", "middle", "end", "throughout")


def frame_indices(bound: Optional[Tuple[float, float]], fps: float, max_frame: int, num_segments: int,
                  first_idx: int = 0) -> np.ndarray:
    """Midpoint sampling inside [start, end] seconds: tasks/test_recursive_grounding.py:350-362 (get_index)."""
    start, end = (bound[0], bound[1]) if bound else (-100000, 100000)
    start_idx = max(first_idx, round(start * fps))
    end_idx = min(round(end * fps), max_frame)
    seg = float(end_idx - start_idx) / num_segments
    idx = np.arange(num_segments)
    # int(start_idx + seg / 2 + np.round(seg * idx)): truncation towards zero, like the reference's int()
    return np.trunc(start_idx + seg / 2 + np.round(seg * idx)).astype(np.int64)


def narrow(bound: Tuple[float, float], answer: str) -> Tuple[float, float]:
    """The segment an answer selects: tasks/test_recursive_grounding.py:456-464."""
    mid, quarter = (bound[1] + bound[0]) / 2, (bound[1] - bound[0]) / 4
    if answer == "beginning":
        return (bound[0], mid)
    if answer == "end":
        return (mid, bound[1])
    if answer == "middle":
        return (mid - quarter, mid + quarter)
    return (bound[0], bound[1])          # 'throughout'


class FrameSource:
    """What the batcher needs from the video side (decord + GroupScale + GroupCenterCrop in the reference,
    :364-385, :333-336).  info(example) -> (fps, number of frames); frames(example, indices, out) writes decoded,
    resized, cropped frames uint8 [T,H,W,3] into `out` (a numpy view of a pinned buffer)."""

    def info(self, example) -> Tuple[float, int]:
        raise NotImplementedError

    def frames(self, example, indices: np.ndarray, out: np.ndarray) -> None:
        raise NotImplementedError


class SyntheticFrameSource(FrameSource):
    """Deterministic stand-in for a video corpus: frame f of example e is a fixed pseudo-random image (a per-example
    base pattern plus a per-frame shift), so any two readers of the same (e, f) see the same bytes."""

    def __init__(self, size: int = 224, fps: float = 3.0, seed: int = 0):
        self.size, self.fps, self.seed = size, fps, seed
        self._base: Dict[int, np.ndarray] = {}

    def n_frames(self, example: int) -> int:
        return 90 + (example * 37) % 120             # 30-70 s at 3 fps, like Charades clips

    def info(self, example):
        return self.fps, self.n_frames(example)

    def frames(self, example, indices, out):
        base = self._base.get(example)
        if base is None:
            rng = np.random.Generator(np.random.PCG64([self.seed, int(example)]))
            base = rng.integers(0, 256, size=(self.size, self.size, 3), dtype=np.uint8)
            if len(self._base) > 256:
                self._base.clear()
            self._base[example] = base
        for t, f in enumerate(indices):
            f = int(f)
            np.add(np.roll(base, f % self.size, axis=1), np.uint8((f * 7) & 0xFF), out=out[t])   # wraps mod 256


class RecursiveGroundingBatcher:
    def __init__(self, encode: Callable, source: FrameSource, answer_fn: Callable, instruction, *, max_turns: int = 4,
                 num_segments: int = 12, micro_batch: int = 32, size: int = 224, device="cuda", keep_outputs: bool = False):
        """encode(frames_u8 [B,T,H,W,3] on `device`, instruction) -> (inputs_llama [B,96,4096], use_image): normally
        `model.encode_frames`.  answer_fn(example, turn, inputs_llama_row [96,4096], bound) -> one of CHOICES: stands in
        for infer_mvbench + the reply matching of :444-454.  instruction: what every turn passes as `system`
        (:270-272): a string (tokenised once per micro-batch by the model's tokenizer) or a pre-tokenised
        (input_ids [1,Lt], attention_mask [1,Lt]) pair, broadcast over the micro-batch."""
        self.encode, self.source, self.answer_fn, self.instruction = encode, source, answer_fn, instruction
        self.max_turns, self.T, self.mb, self.size = max_turns, num_segments, micro_batch, size
        self.device = torch.device(device)
        self.keep_outputs = keep_outputs
        self.cuda = self.device.type == "cuda"
        shape = (micro_batch, num_segments, size, size, 3)
        # two pinned staging buffers + two device buffers: decode / copy of micro-batch k+1 overlaps the encode of k
        self.host = [torch.empty(shape, dtype=torch.uint8, pin_memory=self.cuda) for _ in range(2)]
        self.dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(2)] if self.cuda else self.host
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.copied = [torch.cuda.Event() for _ in range(2)] if self.cuda else None     # H2D of buffer i finished
        self.consumed = [torch.cuda.Event() for _ in range(2)] if self.cuda else None   # encode read device buffer i
        self.stats = {"encode_calls": 0, "clips": 0, "waves": 0, "h2d_bytes": 0}

    def _instruction_for(self, b: int):
        ins = self.instruction
        if isinstance(ins, (tuple, list)) and len(ins) == 2 and torch.is_tensor(ins[0]):
            return (ins[0].expand(b, -1), ins[1].expand(b, -1))
        return [ins] * b if isinstance(ins, str) else ins

    def _stage(self, slot: int, chunk: List[dict]):
        """Host side of one micro-batch: decode into the pinned buffer, then the async H2D on the copy stream."""
        if self.cuda:
            self.copied[slot].synchronize()      # the last H2D out of this pinned buffer has left it
        buf = self.host[slot].numpy()
        for j, st in enumerate(chunk):
            fps, nfr = st["info"]
            idx = frame_indices(st["bound"], fps, nfr - 1, self.T)
            self.source.frames(st["example"], idx, buf[j])
        b = len(chunk)
        if self.cuda:
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.consumed[slot])      # the previous user of this device buffer is done
                self.dev[slot][:b].copy_(self.host[slot][:b], non_blocking=True)
                self.copied[slot].record(self.copy_stream)
        self.stats["h2d_bytes"] += b * self.host[slot][0].numel()

    def run(self, examples: Sequence, rank: int = 0, world: int = 1) -> Dict:
        """Returns {example: {"bound": (start, end), "answers": [...], "duration": seconds[, "outputs": [...]]}} for this
        rank's block of `examples` (inference(), :409-423)."""
        from .dist import shard_range
        lo, hi = shard_range(len(examples), rank, world)
        states = []
        for ex in list(examples)[lo:hi]:
            fps, nfr = self.source.info(ex)
            states.append({"example": ex, "info": (fps, nfr), "bound": None, "duration": nfr / fps, "answers": [],
                           "outputs": [], "done": False})
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            for e in self.consumed:
                e.record(cur)
        for turn in range(self.max_turns):
            active = [s for s in states if not s["done"]]
            if not active:
                break
            self.stats["waves"] += 1
            chunks = [active[i:i + self.mb] for i in range(0, len(active), self.mb)]
            self._stage(0, chunks[0])
            for k, chunk in enumerate(chunks):
                slot, b = k & 1, len(chunk)
                if self.cuda:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(self.copied[slot])
                out, _ = self.encode(self.dev[slot][:b], self._instruction_for(b))
                if self.cuda:
                    self.consumed[slot].record(torch.cuda.current_stream(self.device))
                self.stats["encode_calls"] += 1
                self.stats["clips"] += b
                if k + 1 < len(chunks):
                    # host decode + H2D of the next micro-batch run while the GPU works on this one: the encode above
                    # only enqueued launches, nothing has waited for it yet
                    self._stage((k + 1) & 1, chunks[k + 1])
                # the LLM's turn (host-visible decision): needs this micro-batch's result
                for j, st in enumerate(chunk):
                    if st["bound"] is None:
                        st["bound"] = (0, st["duration"])                      # :431-432
                    ans = self.answer_fn(st["example"], turn, out[j], st["bound"])
                    if ans not in CHOICES:
                        raise RuntimeError(f"answer_fn returned {ans!r}; expected one of {CHOICES}")
                    st["answers"].append(ans)
                    if self.keep_outputs:
                        st["outputs"].append(out[j].detach().float().cpu())
                    st["bound"] = narrow(st["bound"], ans)
                    if ans == "throughout":                                    # :421-422
                        st["done"] = True
        res = {}
        for s in states:
            r = {"bound": s["bound"], "answers": s["answers"], "duration": s["duration"]}
            if self.keep_outputs:
                r["outputs"] = s["outputs"]
            res[s["example"]] = r
        return res
