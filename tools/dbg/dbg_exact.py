import contextlib, io, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from models import build_vae_var
from var_amd.detinit import fill_module_device_
z, meta = util.load_case('t_pn12345')
with contextlib.redirect_stdout(io.StringIO()):
    vae, var = build_vae_var(device='cuda', patch_nums=tuple(meta['patch_nums']), depth=meta['depth'], ch=meta['ch'])
fill_module_device_(var, meta['depth'], 0, 'var.'); fill_module_device_(vae, meta['depth'], 0, 'vae.')
var.eval(); vae.eval()
gt = torch.from_numpy(z['idx'].astype(np.int64)).cuda()
labels = torch.tensor(meta['labels'], device='cuda')
eng = var.engine()
full = eng.sample(2, labels, None, 1.5, 0, 0.0, gt_tokens=gt, keep_mask=torch.ones_like(gt, dtype=torch.bool), trace=True).clone()
fh_eng = eng.last_trace['f_hat'][-1].clone()
ms, cur = [], 0
for pn in meta['patch_nums']:
    ms.append(gt[:, cur:cur + pn * pn].contiguous()); cur += pn * pn
fh2 = vae.quantize.hip_engine().fhat_from_scales(ms, tuple(meta['patch_nums']), from_tokens=True, last_one=True)
print('f_hat equal', torch.equal(fh_eng, fh2), float((fh_eng - fh2).abs().max()))
nhwc = fh2.permute(0, 2, 3, 1).contiguous()
d1 = vae._decoder_engine().decode_nhwc(nhwc, denorm=True).clone()
d1b = vae._decoder_engine().decode_nhwc(nhwc, denorm=True).clone()
d2 = vae._decoder_engine().decode_nhwc(nhwc, denorm=False).clone().add_(1).mul_(0.5)
print('decode twice equal', torch.equal(d1, d1b), 'denorm vs torch', torch.equal(d1, d2), float((d1 - d2).abs().max()), 'engine img vs d1', torch.equal(full, d1), float((full - d1).abs().max()))
img2 = vae.fhat_to_img(fh2).add_(1).mul_(0.5)
print('fhat_to_img', torch.equal(img2, d1), float((img2 - d1).abs().max()))
# f16 batch-slice
z, meta = util.load_case('t_pn12345')
V = var.V
g = torch.Generator().manual_seed(5)
noise = [torch.empty(4 * pn * pn, V).exponential_(1, generator=g) for pn in var.patch_nums]
lab = torch.tensor([1, 22, 333, 980], device='cuda')
var.set_hip_precision('f16')
a = eng.sample(4, lab, None, 1.5, 900, 0.96, noises=noise, trace=True, decode=False).clone(); tra = {k: [t.clone() if t is not None else None for t in v] for k, v in eng.last_trace.items()}
sub = eng.sample(2, lab[1:3], None, 1.5, 900, 0.96, noises=[n.view(4, -1, V)[1:3].reshape(-1, V) for n in noise], trace=True, decode=False); trs = eng.last_trace
for si in range(len(var.patch_nums)):
    la, ls = tra['logits'][si], trs['logits'][si]
    B = 4
    la_sel = torch.cat([la[1:3], la[B + 1:B + 3]])
    print('f16 scale', si, 'logits equal', torch.equal(la_sel, ls), float((la_sel - ls).abs().max()), 'idx equal', torch.equal(tra['idx'][si][1:3], trs['idx'][si]))
var.set_hip_precision('f32')
