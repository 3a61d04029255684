"""Resolution / aspect-ratio / duration vocabulary of ``OpenSoraPipeline.generate`` — host mirror of
videosys/pipelines/open_sora/data_process.py (ASPECT_RATIO_MAP :40-58, the per-resolution size tables :61-455, ASPECT_RATIOS
:462-476, get_image_size :479-483, NUM_FRAMES_MAP / get_num_frames :486-505, prepare_multi_resolution_info :791-805).

The reference stores 13 literal tables.  Here the sizes are COMPUTED: a named resolution is a pixel budget S, an aspect ratio
"h:w" a number r, and the frame is  H = sqrt(S r), W = H / r  rounded to even integers (144p: truncated) — which is how the
reference's "p" tables were generated ("computed from above code", :61).  The handful of entries where the reference's literal
differs from the rule by one rounding step are listed in ``_LITERAL`` so every (resolution, aspect_ratio) pair returns exactly
the reference's size; tests/test_host_cpu.py pins all 221 pairs against tests/golden/opensora_image_sizes.json (minted from the
reference file by oracle/make_golden_geometry.py) and against the live reference when it is present.

The PixArt-style bucket tables ("256", "512", "1024", "2048", "2880") are keyed by one-decimal ratios ("1.0", "0.5"), so through
``get_image_size`` only "12:25" ("0.48") resolves on them and every other aspect ratio fails the reference's assert — in
particular ("512", "1:1"): BASELINE's 512x512 cannot be named through the reference's own vocabulary, which is why
``generate(height=, width=)`` exists here as an extension.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

IMG_FPS = 120  # data_process.py:25

# "h:w" names accepted by generate(aspect_ratio=...) (data_process.py:40-58).  The reference files each under the 2-decimal
# string of its ratio; "50:27" is filed under "2.08" (= 25:12, the transpose of "12:25"), so its sizes follow 25:12.
ASPECT_RATIO_NAMES = ("3:8", "9:21", "12:25", "1:2", "9:17", "27:50", "9:16", "5:8", "2:3", "3:4", "1:1", "4:3", "3:2", "16:9",
                      "17:9", "2:1", "50:27")
_RATIO_OVERRIDE = {"50:27": (25, 12)}

# pixel budget of every named resolution (data_process.py:462-476)
PIXELS = {"144p": 256 * 144, "240p": 426 * 240, "360p": 640 * 360, "480p": 854 * 480, "720p": 1280 * 720, "1080p": 1920 * 1080,
          "2k": 2560 * 1440, "4k": 3840 * 2160, "256": 256 * 256, "512": 512 * 512, "1024": 1024 * 1024, "2048": 2048 * 2048,
          "2880": 2880 * 2880}
_BUCKET_TABLES = ("256", "512", "1024", "2048", "2880")

# entries of the reference's literal tables that are one rounding step away from the rule
_LITERAL = {
    ("144p", "27:50"): (141, 260), ("240p", "27:50"): (236, 436), ("240p", "2:3"): (262, 393), ("480p", "12:25"): (444, 925),
    ("480p", "4:3"): (740, 555), ("480p", "2:1"): (906, 454), ("720p", "27:50"): (706, 1306), ("1080p", "27:50"): (1058, 1958),
    ("2k", "27:50"): (1412, 2612), ("4k", "27:50"): (2118, 3918),
}


def ratio_key(aspect_ratio: str) -> str:
    """ASPECT_RATIO_MAP[aspect_ratio]: the 2-decimal string the reference files the ratio under."""
    if aspect_ratio not in ASPECT_RATIO_NAMES:
        raise KeyError(aspect_ratio)
    h, w = _RATIO_OVERRIDE.get(aspect_ratio, tuple(int(v) for v in aspect_ratio.split(":")))
    return f"{h / w:.2f}"


def _even(v: float) -> int:
    return 2 * round(v / 2)


def get_image_size(resolution: str, ar_ratio: str) -> Tuple[int, int]:
    """data_process.py:479-483 — (height, width) of a named resolution at a named aspect ratio; KeyError for unknown names and
    AssertionError (same message) where the reference's table has no such entry."""
    key = ratio_key(ar_ratio)
    if resolution not in PIXELS:
        raise KeyError(resolution)
    S = PIXELS[resolution]
    if resolution in _BUCKET_TABLES:
        # 32-pixel bucket tables keyed "0.48", "0.5", "1.0", ...: only the two-decimal key "0.48" exists in both vocabularies,
        # and its bucket is (11, 23) * side/16 for the power-of-two sides (2880 has no such bucket)
        side = math.isqrt(S)
        assert key == "0.48" and side & (side - 1) == 0, f"Aspect ratio {ar_ratio} not found for resolution {resolution}"
        return (11 * side // 16, 23 * side // 16)
    if (resolution, ar_ratio) in _LITERAL:
        return _LITERAL[(resolution, ar_ratio)]
    h, w = _RATIO_OVERRIDE.get(ar_ratio, tuple(int(v) for v in ar_ratio.split(":")))
    r = h / w
    if resolution == "144p":
        H = int(math.sqrt(S * r))
        return (H, int(H / r))
    H = _even(math.sqrt(S * r))
    return (H, _even(H / r))


def get_num_frames(num_frames) -> int:
    """data_process.py:486-505: "2s" -> 51, "4s" -> 102, ... (17 frames x 3 per second-pair), "1x".."16x" the same ladder."""
    if isinstance(num_frames, str) and len(num_frames) >= 2 and num_frames[-1] in "sx" and num_frames[:-1].isdigit():
        n = int(num_frames[:-1])
        if num_frames[-1] == "s":
            n, rem = divmod(n, 2)
            if rem:
                n = 0
        if n in (1, 2, 4, 8, 16):
            return 51 * n
    return int(num_frames)


def prepare_multi_resolution_info(batch_size: int, image_size, num_frames: int, fps, dtype=torch.bfloat16, device="cpu"):
    """data_process.py:798-805 ("OpenSora" branch): the per-sample conditioning scalars AS TENSORS OF THE MODEL DTYPE — the
    reference builds them in bf16, so e.g. width 854 reaches timestep_transform and the position-embedding scale as 856."""
    fps = fps if num_frames > 1 else IMG_FPS
    mk = lambda v: torch.tensor([v], device=device, dtype=dtype).repeat(batch_size)
    return dict(height=mk(image_size[0]), width=mk(image_size[1]), num_frames=mk(num_frames), ar=mk(image_size[0] / image_size[1]),
                fps=mk(fps))
