"""Fixed sinusoid position tables of the UMT encoder.

Behaviour follows models/blip2/vit.py:163-194 (get_sinusoid_encoding_table):
a float64 sin/cos table for `ckpt_num_frame * L` positions, cast to fp32,
then -- when the run uses a different number of frames -- linearly
interpolated along T per (h, w, c) with align_corners=False (what
F.interpolate(mode='linear') does, vit.py:186).  The reference builds the table
with nested Python list comprehensions (seconds); this is a vectorised
restatement that produces the same fp32 values.

Rounding of the interpolation: torch's CPU upsample_linear1d kernel is compiled with
a*b+c contraction, i.e. it evaluates  src = fma(scale, i + 0.5, -0.5)  and
out = fma(l0, x[i0], l1 * x[i1]);  with exactly these two fused steps the table equals
the reference's (tests/golden/pos_embed.npz) bit for bit at T = 8, 12, 16.  The fused
steps are emulated in float64 (the fp32 product is exact there).
"""
import numpy as np
import torch


def _angle_table(n_position: int, d_hid: int) -> torch.Tensor:
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid, dtype=np.int64)[None, :]
    # vit.py:167  position / np.power(10000, 2 * (hid_j // 2) / d_hid)
    denom = np.power(10000.0, 2.0 * (j // 2).astype(np.float64) / float(d_hid))
    tab = pos / denom
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.tensor(tab, dtype=torch.float32)


def _lerp_time(x: torch.Tensor, new_t: int) -> torch.Tensor:
    """x [T, L, C] fp32 -> [new_t, L, C]; upsample_linear1d, align_corners=False."""
    T = x.shape[0]
    scale = np.float32(T) / np.float32(new_t)
    out = torch.empty((new_t,) + tuple(x.shape[1:]), dtype=torch.float32)
    for i in range(new_t):
        src = np.float32(np.float64(scale) * np.float64(i + 0.5) - 0.5)        # fma(scale, i + 0.5, -0.5)
        if src < 0:
            src = np.float32(0.0)
        i0 = int(src)
        i1 = i0 + 1 if i0 < T - 1 else i0
        l1 = np.float32(src - np.float32(i0))
        l0 = np.float32(1.0) - l1
        out[i] = (float(l0) * x[i0].double() + (float(l1) * x[i1]).double()).float()   # fma(l0, x0, l1 * x1)
    return out


def video_pos_embed(num_frames: int, patches_per_frame: int, d_hid: int, ckpt_num_frame: int = 4) -> torch.Tensor:
    """[1, num_frames*L, d] fp32 -- encoder.pos_embed (vit.py:278)."""
    L = patches_per_frame
    if ckpt_num_frame != -1 and ckpt_num_frame != num_frames:
        tab = _angle_table(ckpt_num_frame * L, d_hid).reshape(ckpt_num_frame, L, d_hid)
        return _lerp_time(tab, num_frames).reshape(1, num_frames * L, d_hid)
    return _angle_table(num_frames * L, d_hid).unsqueeze(0)


def image_pos_embed(patches_per_frame: int, d_hid: int) -> torch.Tensor:
    """[1, L, d] fp32 -- encoder.img_pos_embed (vit.py:279)."""
    return _angle_table(patches_per_frame, d_hid).unsqueeze(0)
