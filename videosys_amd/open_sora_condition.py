"""Image / video conditioning of the Open-Sora pipeline — host logic either side of the denoise step.

Mirror of the module-level helpers of videosys/pipelines/open_sora/pipeline_open_sora.py: ``extract_json_from_prompts``
(:719-733), ``collect_references_batch`` (:736-750), ``parse_mask_strategy`` / ``find_nearest_point`` /
``apply_mask_strategy`` (:795-854), ``append_generated`` (:857-871) and ``dframe_to_frame`` (:874-876).  Same names, argument
meaning and error behaviour; tests/test_host_cpu.py checks them against values minted from the reference
(tests/golden/stdit3_xmask_small.pt, oracle/make_golden_xmask.py).

A mask strategy is a ';'-separated list of up to six comma-separated fields

    loop_id, ref_id, ref_start, target_start, length, edit_ratio        (defaults 0, 0, 0, 0, 1, 0)

"in loop ``loop_id`` copy ``length`` latent frames of reference ``ref_id`` starting at ``ref_start`` over the frames of z
starting at ``target_start``, and give those frames the mask value ``edit_ratio``" — 0 holds the frames for the whole sampling,
a value in (0, 1) lets them join the denoising once t <= edit_ratio * 1000 (rflow.RFLOW.sample).  Negative starts count from
the end; with ``align`` both starts snap to the nearest multiple of it that still leaves one aligned block.

Everything here is tiny host-side indexing on latents; the pixel <-> latent conversions are the VAE's (vae_open_sora.py).
"""
from __future__ import annotations

import json
import os
import re
from typing import List, Optional, Sequence, Tuple

import torch

_FIELD_DEFAULTS = (0, 0, 0, 0, 1, 0.0)
IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")
VID_EXTENSIONS = (".mp4", ".avi", ".mov", ".mkv")


def parse_mask_strategy(mask_strategy: Optional[str]) -> List[list]:
    """"0,0,-5,0,5,0.3;1" -> [[0, 0, -5, 0, 5, 0.3], [1, 0, 0, 0, 1, 0.0]]; "" / None -> []."""
    if not mask_strategy:
        return []
    groups = []
    for item in mask_strategy.split(";"):
        fields = item.split(",")
        assert 1 <= len(fields) <= 6, f"Invalid mask strategy: {item}"
        vals = [int(v) for v in fields[:5]] + [float(v) for v in fields[5:]]
        groups.append(vals + list(_FIELD_DEFAULTS[len(vals):]))
    return groups


def find_nearest_point(value: int, point: int, max_value: int) -> int:
    """Snap ``value`` to a multiple of ``point``: up when it is past the middle of its block and a whole block still fits
    under ``max_value``, down otherwise."""
    block = value // point
    if value % point > point / 2 and block < max_value // point - 1:
        block += 1
    return block * point


def apply_mask_strategy(z: torch.Tensor, refs_x: Sequence[Sequence[torch.Tensor]], mask_strategys: Sequence[Optional[str]],
                        loop_i: int, align: Optional[int] = None) -> Optional[torch.Tensor]:
    """Paste the reference latents into ``z`` [B, C, T, H, W] IN PLACE and return the per-frame mask [B, T] (float, 1 = generate)
    for loop ``loop_i``; None when there is no strategy list at all."""
    if len(mask_strategys) == 0:
        return None
    T = z.shape[2]
    masks = torch.ones(len(mask_strategys), T, dtype=torch.float, device=z.device)
    for b, strategy in enumerate(mask_strategys):
        for loop_id, ref_id, ref_start, target_start, length, edit_ratio in parse_mask_strategy(strategy):
            if loop_id != loop_i:
                continue
            ref = refs_x[b][ref_id]                       # [C, T_ref, H, W]
            T_ref = ref.shape[1]
            if ref_start < 0:
                ref_start += T_ref
            if target_start < 0:
                target_start += T
            if align is not None:
                ref_start = find_nearest_point(ref_start, align, T_ref)
                target_start = find_nearest_point(target_start, align, T)
            n = min(length, T - target_start, T_ref - ref_start)
            z[b, :, target_start:target_start + n] = ref[:, ref_start:ref_start + n].to(device=z.device, dtype=z.dtype)
            masks[b, target_start:target_start + n] = edit_ratio
    return masks


def append_generated(vae_encode, generated_video: torch.Tensor, refs_x: List[Optional[list]], mask_strategy: List[Optional[str]],
                     loop_i: int, condition_frame_length: int, condition_frame_edit: float) -> Tuple[list, list]:
    """After loop ``loop_i - 1``: encode the clip just generated, append it to every sample's reference list and add the strategy
    "in loop loop_i, start from the last ``condition_frame_length`` latent frames of that clip" (the continuation rule)."""
    ref_x = vae_encode(generated_video)
    for j in range(len(refs_x)):
        if refs_x[j] is None:
            refs_x[j] = [ref_x[j]]
        else:
            refs_x[j].append(ref_x[j])
        head = (mask_strategy[j] + ";") if mask_strategy[j] else ""
        mask_strategy[j] = head + f"{loop_i},{len(refs_x[j]) - 1},-{condition_frame_length},0,{condition_frame_length},{condition_frame_edit}"
    return refs_x, mask_strategy


def dframe_to_frame(num: int) -> int:
    """Latent frames -> pixel frames (5 latent frames per 17-frame micro batch)."""
    assert num % 5 == 0, f"Invalid num: {num}"
    return num // 5 * 17


def extract_json_from_prompts(prompts: List[str], reference: list, mask_strategy: list):
    """Split an optional JSON tail ``{"reference_path": ..., "mask_strategy": ...}`` off every prompt; its values override the
    entries of ``reference`` / ``mask_strategy`` (modified in place and returned)."""
    texts = []
    for i, prompt in enumerate(prompts):
        parts = re.split(r"(?=[{])", prompt)
        assert len(parts) <= 2, f"Invalid prompt: {prompt}"
        texts.append(parts[0])
        if len(parts) == 2:
            for key, val in json.loads(parts[1]).items():
                assert key in ("reference_path", "mask_strategy"), f"Invalid key: {key}"
                if key == "reference_path":
                    reference[i] = val
                else:
                    mask_strategy[i] = val
    return texts, reference, mask_strategy


def _resize_crop_to_fill(img, size):
    """data_process.py resize_crop_to_fill: scale so the image covers (h, w), centre-crop the overflow."""
    from PIL import Image

    th, tw = size
    w, h = img.size
    rh, rw = th / h, tw / w
    if rh > rw:
        sh, sw = th, round(w * rh)
        img = img.resize((sw, sh), Image.BICUBIC)
        left = int(round((sw - tw) / 2.0))
        return img.crop((left, 0, left + tw, th))
    sh, sw = round(h * rw), tw
    img = img.resize((sw, sh), Image.BICUBIC)
    top = int(round((sh - th) / 2.0))
    return img.crop((0, top, tw, top + th))


def read_from_path(path: str, image_size, transform_name: str = "resize_crop") -> torch.Tensor:
    """An image file -> [3, 1, H, W], a video file -> [3, T, H, W], in [-1, 1] (data_process.py:761-788 with the "resize_crop"
    transform: cover + centre crop, ToTensor, Normalize(0.5, 0.5)).  Images are read with PIL.  The reference decodes videos with
    torchvision.io / av, which this image does not have: Motion-JPEG AVI files (what ``save_video`` writes here, utils.read_mjpeg_avi)
    are decoded, any other video must be handed over as a tensor (``refs=[tensor]``).  (The reference's video branch applies
    ``resize_crop_to_fill`` — a PIL function, data_process.py:742-758 — to the tensor clip and cannot run as written; the per-frame
    cover + crop it intends is what is done here.)"""
    ext = os.path.splitext(path)[-1].lower()
    assert ext in VID_EXTENSIONS or ext in IMG_EXTENSIONS, f"Unsupported file format: {ext}"
    assert transform_name == "resize_crop", transform_name
    import numpy as np
    from PIL import Image

    if ext in VID_EXTENSIONS:
        from .utils import read_mjpeg_avi

        try:
            frames = read_mjpeg_avi(path) if ext == ".avi" else None
        except ValueError:
            frames = None
        if not frames:
            raise NotImplementedError(f"{path}: no decoder for this video in this environment (Motion-JPEG .avi files are read); pass "
                                      "the reference clip as a [3, T, H, W] tensor in [-1, 1] (refs=[tensor]) or as latents")
    else:
        with open(path, "rb") as fh:
            frames = [Image.open(fh).convert("RGB")]
    arr = np.stack([np.asarray(_resize_crop_to_fill(f, tuple(image_size)), dtype=np.uint8) for f in frames])     # [T, H, W, 3]
    x = torch.from_numpy(arr.copy()).permute(3, 0, 1, 2).float().div_(255.0)
    return x.sub_(0.5).div_(0.5)


def collect_references_batch(reference_paths: Sequence, vae_encode, image_size) -> List[list]:
    """Per sample: '' -> []; 'a.png;b.png' -> the encoded latents [C, T, H, W] of every file.  Besides the reference's path
    strings an entry may be a list whose items are paths, pixel tensors [3, T, H, W] in [-1, 1] (encoded) or latents
    [4, T, h, w] (taken as they are)."""
    out = []
    for entry in reference_paths:
        if entry is None or (isinstance(entry, str) and entry == ""):
            out.append([])
            continue
        items = entry.split(";") if isinstance(entry, str) else list(entry)
        refs = []
        for it in items:
            if torch.is_tensor(it) and it.dim() == 4 and it.shape[0] != 3:
                refs.append(it)
                continue
            pix = it if torch.is_tensor(it) else read_from_path(it, image_size, transform_name="resize_crop")
            if vae_encode is None:
                raise RuntimeError("a pixel-space reference needs the VAE encoder: attach a VAE to the pipeline or pass latents")
            refs.append(vae_encode(pix.unsqueeze(0)).squeeze(0))
        out.append(refs)
    return out
