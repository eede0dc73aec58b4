import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from var_amd import hip
torch.manual_seed(0)
B2,H,Lmax=8,16,680
kc=torch.randn(B2,H,Lmax,64,device='cuda').half(); vc=torch.randn(B2,H,Lmax,64,device='cuda').half()
l,cur=256,680
q=torch.randn(B2*l,H*64,device='cuda').half(); outs=[]
for r in range(10):
    out=torch.empty_like(q); hip.call('attn_cached_f16',q,kc,vc,out,B2,l,H,cur,Lmax); outs.append(out.clone())
bad=[i for i in range(1,10) if not torch.equal(outs[i],outs[0])]
print('dbg', os.environ.get('VARHIP_ATTN16_DBG'), 'mismatching', bad)
if bad:
    d=(outs[bad[0]]!=outs[0]).nonzero()
    print('ndiff', d.shape[0], 'nan count', int(torch.isnan(outs[0].float()).sum()), 'first diffs (row, col):', d[:12].tolist())
    rows=d[:,0]; print('rows mod 256 (query idx):', sorted(set((rows%256).tolist()))[:40], 'cols mod 64:', sorted(set((d[:,1]%64).tolist()))[:64])
