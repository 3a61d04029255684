"""debug helper: run OpenSoraVAE.encode with a sync + marker after every op wrapper call"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videosys_amd import ops
from videosys_amd.vae_open_sora import OpenSoraVAE, synth_state_dict

hw, frames = int(sys.argv[1]), int(sys.argv[2])
names = ["vae_first_im2col", "gemm128", "conv", "group_norm", "regrid", "subsample", "extract_planar", "softmax_rows"]
for n in names:
    f = getattr(ops, n)
    def wrap(*a, _f=f, _n=n, **k):
        shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)][:3]
        print("->", _n, shapes, flush=True)
        r = _f(*a, **k)
        torch.cuda.synchronize()
        return r
    setattr(ops, n, wrap)
dev = torch.device("cuda:0")
vae = OpenSoraVAE(synth_state_dict(0, encoder=True), device=dev)
x = (torch.rand(1, 3, frames, hw * 8, hw * 8, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
z = vae.encode(x, noise_fn=lambda s: torch.zeros(s))
torch.cuda.synchronize()
print("ok", tuple(z.shape), bool(torch.isfinite(z).all()))
for rep in range(2):
    z2 = vae.encode(x, noise_fn=lambda s: torch.zeros(s))
    torch.cuda.synchronize()
    print("again", rep, bool(torch.equal(z, z2)), flush=True)
