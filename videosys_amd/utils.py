"""videosys/utils/utils.py mirror: set_seed (:19-34), batch_func (:37-52), save_video."""
import logging
import random

import numpy as np
import torch
import torch.distributed as dist


def set_seed(seed, dp_rank=None):
    if seed == -1:
        seed = random.randint(0, 1000000)
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([seed], dtype=torch.int64)
        if torch.cuda.is_available() and dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, 0)
        seed = int(t.item())
    if dp_rank is not None:
        seed = seed + dp_rank
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def str_to_dtype(x: str):
    """utils/utils.py:39-47."""
    table = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
    if x not in table:
        raise RuntimeError(f"Only fp32, fp16 and bf16 are supported, but got {x}")
    return table[x]


def requires_grad(model, flag: bool = True) -> None:
    """utils/utils.py:12-17, for torch modules a caller attaches; the model objects of this build hold plain HBM tables."""
    for p in model.parameters():
        p.requires_grad = flag


def all_exists(paths):
    """utils/utils.py:80-81."""
    import os

    return all(os.path.exists(path) for path in paths)


def empty_cache(func):
    """utils/test.py:6-13: the decorator of the reference's pipeline tests — release the caching allocator before the test."""
    import functools

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        return func(*args, **kwargs)

    return wrapper


def init_logger(logging_dir: str = None, master_only: bool = True):
    """utils/logging.py:6-38: root logger at INFO with the reference's line format (+ ``<logging_dir>/log.txt``); silent on the
    ranks other than 0 when ``master_only``."""
    import logging

    logger = logging.getLogger()
    logger.handlers.clear()
    if dist.is_initialized() and master_only and dist.get_rank() != 0:
        logger.addHandler(logging.NullHandler())
        return logger
    extra = {}
    if logging_dir is not None:
        extra["handlers"] = [logging.StreamHandler(), logging.FileHandler(f"{logging_dir}/log.txt")]
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] [%(levelname)s] [%(filename)s:%(lineno)d:%(funcName)s]   %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S", **extra)
    return logging.getLogger()


def progress_wrap(iterable, enabled: bool = True, **tqdm_kwargs):
    """The denoise loops' progress bar: tqdm on rank 0 when asked for (scheduling_rflow_open_sora.py:219,
    pipeline_latte.py:834), the plain iterable otherwise (other ranks, ``verbose=False``, tqdm not installed)."""
    if not enabled or (dist.is_initialized() and dist.get_rank() != 0):
        return iterable
    try:
        from tqdm import tqdm
    except ImportError:
        return iterable
    return tqdm(iterable, **tqdm_kwargs)


def randn_tensor(shape, generator=None, dtype=None):
    """Start noise the way the pipelines draw it (diffusers' ``randn_tensor``, third-party: drawn on the generator's device, then
    moved by the caller): fp32 from ``generator`` — a torch.Generator, or a list of them, one per sample — else from torch's
    global CPU stream (which set_seed seeded).  The bits do not depend on the box's GPU."""
    if isinstance(generator, (list, tuple)):
        z = torch.cat([torch.randn((1,) + tuple(shape[1:]), generator=g, dtype=torch.float32, device=g.device).cpu() for g in generator], 0)
    elif generator is not None:
        z = torch.randn(tuple(shape), generator=generator, dtype=torch.float32, device=generator.device).cpu()
    else:
        z = torch.randn(tuple(shape), dtype=torch.float32)
    return z if dtype is None else z.to(dtype)


def check_prompt_args(prompt, negative_prompt, prompt_embeds=None, negative_prompt_embeds=None):
    """The prompt / embedding combinations the Latte and CogVideoX pipelines refuse in ``check_inputs``
    (pipelines/latte/pipeline_latte.py:484-516, pipelines/cogvideox/pipeline_cogvideox.py:404-434), same ValueErrors."""
    if prompt is not None and prompt_embeds is not None:
        raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
    if prompt is None and prompt_embeds is None:
        raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
    if prompt is not None and not isinstance(prompt, (str, list)):
        raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
    if prompt is not None and negative_prompt_embeds is not None:
        raise ValueError("Cannot forward both `prompt` and `negative_prompt_embeds`. Please make sure to only forward one of the two.")
    if negative_prompt is not None and negative_prompt_embeds is not None:
        raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`. Please make sure to only forward one of the two.")
    if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
        raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but got:"
                         f" `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")


def read_component(path, subfolder=None):
    """One component of a LOCAL checkpoint directory in the Hugging Face layout the reference's ``from_pretrained`` calls read
    (``<path>/<subfolder>/config.json`` or ``scheduler_config.json`` + ``*.safetensors``, single file or the sharded
    ``...-00001-of-0000N.safetensors`` set) -> (config dict, state dict or None).  ({}, None) when the directory is not there:
    hub ids cannot be fetched on this box, the callers then fall back to their seeded synthetic weights or refuse."""
    import glob
    import json
    import os

    d = os.path.join(path, subfolder) if (isinstance(path, str) and subfolder) else path
    if not (isinstance(d, str) and os.path.isdir(d)):
        return {}, None
    cfg = {}
    for name in ("config.json", "scheduler_config.json"):
        if os.path.isfile(os.path.join(d, name)):
            with open(os.path.join(d, name)) as fh:
                cfg = json.load(fh)
            break
    files = sorted(glob.glob(os.path.join(d, "*.safetensors")))
    if not files:
        return cfg, None
    from safetensors.torch import load_file

    sd = {}
    for f in files:
        sd.update(load_file(f))
    return cfg, sd


def ctor_kwargs(fn, cfg: dict) -> dict:
    """The entries of a checkpoint's config.json that ``fn`` takes as keywords (bookkeeping keys such as ``_class_name`` dropped)."""
    import inspect

    names = set(inspect.signature(fn).parameters) - {"self", "device", "dtype"}
    return {k: v for k, v in cfg.items() if k in names}


def same_tensor(a, b) -> bool:
    """True when ``b`` is ``a`` or a fresh VIEW OBJECT of exactly the same elements (same storage, offset, shape, strides) at the
    same version counter — what a per-step ``y[rows]`` slice produces under CFG parallel.  Meant for caches that HOLD ``a``:
    the strong reference keeps the storage alive, so its address cannot be recycled for another prompt meanwhile."""
    if a is b:
        return True
    if a is None or b is None or not (torch.is_tensor(a) and torch.is_tensor(b)):
        return False
    return (a.dtype == b.dtype and a.device == b.device and a.shape == b.shape and a.stride() == b.stride()
            and a.storage_offset() == b.storage_offset() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and a._version == b._version)


def batch_func(func, *args):
    return tuple(func(a) if isinstance(a, torch.Tensor) and a.shape[0] > 0 else a for a in args)


def write_mjpeg_avi(frames, path, fps=24, quality=90):
    """uint8 frames [T, H, W, 3] -> a Motion-JPEG AVI file (RIFF 'AVI ': one 'vids' / 'MJPG' stream, every frame a JPEG key frame,
    'idx1' index) with nothing but the standard library and PIL for the JPEG compression.  Plays in ffmpeg / VLC / browsers'
    ``<video>`` fallbacks; used when no video encoder package is installed."""
    import io
    import struct

    from PIL import Image

    frames = np.ascontiguousarray(frames)
    if frames.ndim != 4 or frames.shape[-1] != 3 or frames.dtype != np.uint8:
        raise ValueError(f"expected uint8 frames [T, H, W, 3], got {frames.dtype} {tuple(frames.shape)}")
    T, H, W, _ = frames.shape
    jpegs = []
    for f in frames:
        buf = io.BytesIO()
        Image.fromarray(f, "RGB").save(buf, format="JPEG", quality=quality)
        jpegs.append(buf.getvalue())
    chunk = lambda tag, data: tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    lst = lambda tag, data: b"LIST" + struct.pack("<I", len(data) + 4) + tag + data
    biggest = max((len(j) for j in jpegs), default=0)
    fps_num, fps_den = (int(round(fps * 1000)), 1000) if fps != int(fps) else (int(fps), 1)
    avih = struct.pack("<14I", int(round(1e6 * fps_den / fps_num)), biggest * fps_num // fps_den, 0, 0x10, T, 0, 1, biggest, W, H, 0, 0, 0, 0)
    strh = b"vidsMJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, fps_den, fps_num, 0, T, biggest, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi, index, off = b"", b"", 4                      # index offsets count from the 'movi' fourcc
    for j in jpegs:
        index += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        c = chunk(b"00dc", j)
        movi += c
        off += len(c)
    body = b"AVI " + hdrl + lst(b"movi", movi) + chunk(b"idx1", index)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return path


def read_mjpeg_avi(path):
    """The frames of a Motion-JPEG AVI (what write_mjpeg_avi produces; also ffmpeg's ``-c:v mjpeg``) as PIL RGB images, with the
    standard library + PIL only.  Any other codec / container: ValueError."""
    import io
    import struct

    from PIL import Image

    with open(path, "rb") as fh:
        d = fh.read()
    if d[:4] != b"RIFF" or d[8:12] != b"AVI ":
        raise ValueError(f"{path}: not a RIFF AVI file")
    sh = d.find(b"strh")
    if sh < 0 or d[sh + 8:sh + 12] != b"vids" or d[sh + 12:sh + 16].upper() not in (b"MJPG", b"JPEG"):
        raise ValueError(f"{path}: the video stream is not Motion-JPEG (handler {d[sh + 12:sh + 16]!r})")
    pos = d.find(b"movi")
    if pos < 0:
        raise ValueError(f"{path}: no 'movi' list")
    end = pos - 4 + struct.unpack("<I", d[pos - 4:pos])[0]
    pos += 4
    frames = []
    while pos + 8 <= min(end, len(d)):
        tag, size = d[pos:pos + 4], struct.unpack("<I", d[pos + 4:pos + 8])[0]
        if tag == b"LIST":                     # 'rec ' groups: step inside
            pos += 12
            continue
        if tag[2:] in (b"dc", b"db") and size:
            frames.append(Image.open(io.BytesIO(d[pos + 8:pos + 8 + size])).convert("RGB"))
        pos += 8 + size + (size & 1)
    return frames


def save_video(video, output_path, fps=24):
    """utils/utils.py:84-92: uint8 frames [T, H, W, C] -> ``output_path``; rank 0 only in a process group.  The reference hands the
    frames to ``imageio.mimwrite`` (third-party, needs an ffmpeg plugin for .mp4); when imageio is installed that is what runs.
    Without it there is no H.264 encoder to produce an .mp4: the frames are written as a Motion-JPEG AVI next to the requested name
    (``sunset.mp4`` -> ``sunset.avi``, write_mjpeg_avi) and the path actually written is returned."""
    import os

    if dist.is_initialized() and dist.get_rank() != 0:
        return None
    os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
    frames = video.detach().cpu().numpy() if torch.is_tensor(video) else np.asarray(video)
    try:
        import imageio
    except ImportError:
        imageio = None
    if imageio is not None and hasattr(imageio, "mimwrite"):
        imageio.mimwrite(output_path, frames, fps=fps)
        return output_path
    if frames.dtype != np.uint8:   # float frames in [0, 1] (Latte's single-image output)
        frames = (np.clip(frames, 0, 1) * 255).round().astype(np.uint8)
    if frames.ndim == 4 and frames.shape[1] == 3 and frames.shape[-1] != 3:
        frames = frames.transpose(0, 2, 3, 1)          # [T, 3, H, W] -> [T, H, W, 3]
    alt = output_path if output_path.lower().endswith(".avi") else os.path.splitext(output_path)[0] + ".avi"
    if alt != output_path:
        logging.getLogger(__name__).warning("no video encoder package (imageio) in this environment: writing Motion-JPEG %s "
                                            "instead of %s", alt, output_path)
    return write_mjpeg_avi(frames, alt, fps=fps)


class HostOffload:
    """``cpu_offload=True`` (pipeline_open_sora.py:241-244 ``enable_model_cpu_offload``; diffusers hooks in the reference): one
    stage of a pipeline — text encoder, transformer or VAE decoder — keeps its weights in PINNED host memory and holds HBM only
    while it runs.  The kernels here take raw device pointers, so nothing can page in lazily: ``to_device()`` brings every weight
    tensor of the object graph back before the stage is entered and ``to_host()`` releases the HBM afterwards.

    Works on any of this package's weight holders: it walks attributes, dicts, lists, tuples and SimpleNamespaces, moves every CUDA
    tensor it finds and preserves aliasing (one tensor object referenced from two places stays one tensor)."""

    def __init__(self, obj, device=None, skip=("_ws", "_text_cache", "_rope_cache", "_pos_cache", "_fps_cache", "_bias_cache", "_padded",
                                               "_fold")):   # (_fold: a device table of weight ADDRESSES — rebuilt after the weights return)
        self.obj = obj
        self.skip = set(skip)
        self.device = torch.device(device) if device is not None else getattr(obj, "device", torch.device("cuda"))
        self.on_device = True
        self._parked = set()   # ids of the pinned host copies made by to_host(): only these travel back (tensors the holder
                               # keeps on the host on purpose — rope.freqs, bias tables — stay where they are)

    def _walk(self, o, move, memo, depth=0):
        if torch.is_tensor(o):
            key = id(o)
            if key not in memo:
                memo[key] = move(o)
            return memo[key]
        if depth > 8:
            return o
        if isinstance(o, dict):
            for k in list(o):
                if k not in self.skip:
                    o[k] = self._walk(o[k], move, memo, depth + 1)
            return o
        if isinstance(o, list):
            for i in range(len(o)):
                o[i] = self._walk(o[i], move, memo, depth + 1)
            return o
        if isinstance(o, tuple):
            return tuple(self._walk(v, move, memo, depth + 1) for v in o)
        mod = type(o).__module__ or ""
        if mod.startswith("videosys_amd") or mod == "types":   # this package's holders and SimpleNamespace
            d = getattr(o, "__dict__", None)
            if d is not None:
                for k in list(d):
                    if k not in self.skip and k != "device":
                        d[k] = self._walk(d[k], move, memo, depth + 1)
        return o

    def drop_workspaces(self):
        for name in self.skip:
            v = getattr(self.obj, name, None)
            if isinstance(v, dict):
                v.clear()
            elif v is not None and name in ("_text_cache", "_rope_cache") and not isinstance(v, dict):
                setattr(self.obj, name, None)

    def to_host(self):
        if not self.on_device:
            return self
        pin = torch.cuda.is_available()

        parked = self._parked = set()
        keep = self._keep = []   # the ids stay valid while the copies are referenced here

        def move(t):
            if not t.is_cuda:
                return t
            h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=pin)
            h.copy_(t)
            parked.add(id(h))
            keep.append(h)
            return h

        self.drop_workspaces()
        self._walk(self.obj, move, {})
        self.on_device = False
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        return self

    def to_device(self):
        if self.on_device:
            return self
        parked = self._parked
        self._walk(self.obj, lambda t: t.to(self.device, non_blocking=True) if id(t) in parked else t, {})
        self._parked, self._keep = set(), []
        self.on_device = True
        return self


class StagedOffloadMixin:
    """Pipeline side of ``cpu_offload``: ``_init_stages`` parks the named weight holders on the host when the config asks for it,
    ``_enter_stage(name)`` makes exactly that stage resident (``None``: none).  A no-op without cpu_offload."""

    _stages = None

    def _init_stages(self, enabled: bool, device, **holders):
        self._stages = {}
        if enabled:
            for name, obj in holders.items():
                if obj is not None and hasattr(obj, "__dict__"):
                    self._stages[name] = HostOffload(obj, device=device).to_host()

    def _enter_stage(self, name):
        if not self._stages:
            return
        for other, off in self._stages.items():
            if other != name:
                off.to_host()
        if name in self._stages:
            self._stages[name].to_device()
            self._after_onload(name)

    def _after_onload(self, name):   # hook: refresh attribute aliases of weight-table entries
        pass

    def _set_seed(self, seed):
        """The pipelines' ``_set_seed`` (e.g. pipelines/open_sora/pipeline_open_sora.py:253-257): one process seeds with ``seed``,
        a process group with ``seed + dp_rank`` — the ranks of one data-parallel replica draw the same noise, replicas differ.
        -1 draws a fresh seed on rank 0 and broadcasts it.  Returns the seed this rank used."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return set_seed(seed)
        return set_seed(seed, self.transformer.parallel_manager.dp_rank)
