import contextlib, io, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from models import build_vae_var
from var_amd.detinit import fill_module_device_
pns=(1,2,3,4,5,6,8,10,13,16)
with contextlib.redirect_stdout(io.StringIO()):
    vae, var = build_vae_var(device='cuda', patch_nums=pns, depth=16, ch=160)
fill_module_device_(var, 16, 0, 'var.'); fill_module_device_(vae, 16, 0, 'vae.')
var.eval(); vae.eval()
V, B = var.V, 4
g = torch.Generator().manual_seed(5)
noise = [torch.empty(B * pn * pn, V).exponential_(1, generator=g) for pn in pns]
labels = torch.tensor([1, 22, 333, 980], device='cuda')
eng = var.engine()
var.set_hip_precision('f16')
runs=[]
for r in range(4):
    img = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, trace=True).clone()
    tr = {k: [t.clone() if t is not None else None for t in v] for k, v in eng.last_trace.items()}
    runs.append((img, tr))
for r in range(1,4):
    for si in range(len(pns)):
        e = torch.equal(runs[r][1]['logits'][si], runs[0][1]['logits'][si]); ei = torch.equal(runs[r][1]['idx'][si], runs[0][1]['idx'][si])
        if not e or not ei: print('run', r, 'scale', si, 'logits equal', e, 'ndiff', int((runs[r][1]['logits'][si] != runs[0][1]['logits'][si]).sum()), 'maxdiff', float((runs[r][1]['logits'][si] - runs[0][1]['logits'][si]).abs().max()), 'idx equal', ei)
    print('run', r, 'f_hat equal', torch.equal(runs[r][1]['f_hat'][-1], runs[0][1]['f_hat'][-1]), 'img equal', torch.equal(runs[r][0], runs[0][0]), float((runs[r][0]-runs[0][0]).abs().max()))
# decoder alone, repeated
fh = eng.sample(B, labels, None, 1.5, 900, 0.96, noises=noise, decode=False)
nhwc = fh.permute(0,2,3,1).contiguous()
d = [vae._decoder_engine().decode_nhwc(nhwc).clone() for _ in range(4)]
print('decoder16 alone equal:', [torch.equal(d[i], d[0]) for i in range(1,4)], [float((d[i]-d[0]).abs().max()) for i in range(1,4)])
