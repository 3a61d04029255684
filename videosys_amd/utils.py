"""videosys/utils/utils.py mirror: set_seed (:19-34), batch_func (:37-52), save_video."""
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


def batch_func(func, *args):
    return tuple(func(a) if isinstance(a, torch.Tensor) and a.shape[0] > 0 else a for a in args)


def save_video(video, output_path, fps=24):
    """uint8 [T,H,W,C] -> file.  imageio is not installed in this image: frames go to an .npy next to the name."""
    import os

    os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
    np.save(os.path.splitext(output_path)[0] + ".npy", video.cpu().numpy())
