"""experiment: consecutive sampling calls issued on two alternating HIP streams (two model instances, own workspaces) vs one stream"""
import sys, os, time, io, contextlib, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from var_amd import detinit
from models import build_vae_var
ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=8); ap.add_argument('--batch', type=int, default=64); ap.add_argument('--nstreams', type=int, default=2); ap.add_argument('--precs', default='f32,f16')
a = ap.parse_args()
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
pairs = []
for i in range(a.nstreams):
    with contextlib.redirect_stdout(io.StringIO()):
        vae, var = build_vae_var(device=dev, patch_nums=pns, depth=16, ch=160)
    detinit.fill_module_device_(var, 16, 0, 'var.'); detinit.fill_module_device_(vae, 16, 0, 'vae.')
    var.eval(); vae.eval(); var.rng = torch.Generator(device=dev)
    pairs.append((vae, var))
labels = ((torch.arange(a.batch) * 7) % 1000).to(dev)
streams = [torch.cuda.Stream() for _ in range(a.nstreams)]
def run(nstreams, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for i in range(steps):
        k = i % nstreams
        with torch.cuda.stream(streams[k]), torch.inference_mode():
            outs.append(pairs[k][1].autoregressive_infer_cfg(a.batch, labels, g_seed=i, cfg=1.5, top_k=900, top_p=0.96))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, outs
for prec in a.precs.split(','):
    for _, var in pairs: var.set_hip_precision(prec)
    run(a.nstreams, a.nstreams)
    one, o1 = run(1, a.steps)
    two, o2 = run(a.nstreams, a.steps)
    same = all(torch.equal(x, y) for x, y in zip(o1, o2))
    print(f'{prec}: one stream {one:.2f} ms/step, {a.nstreams} streams {two:.2f} ms/step ({one / two:.3f}x), identical images: {same}', flush=True)
