import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from var_amd import hip
torch.manual_seed(0)
B2,H,Lmax=8,16,680
kc=torch.randn(B2,H,Lmax,64,device='cuda').half(); vc=torch.randn(B2,H,Lmax,64,device='cuda').half()
cur=0
for pn in (1,2,3,4,5,6,8,10,13,16):
    l=pn*pn; cur+=l
    q=torch.randn(B2*l,H*64,device='cuda').half(); outs=[]
    for r in range(8):
        out=torch.empty_like(q); hip.call('attn_cached_f16',q,kc,vc,out,B2,l,H,cur,Lmax); outs.append(out.clone())
    bad=[i for i in range(1,8) if not torch.equal(outs[i],outs[0])]
    print(f'attn16 l={l} curL={cur}: mismatching {bad}', (float((outs[bad[0]].float()-outs[0].float()).abs().max()), int((outs[bad[0]]!=outs[0]).sum())) if bad else '')
# qkv epilogue
C=1024;K=1024
for l,pos0 in ((25,30),(36,55),(64,91),(256,424)):
    M=B2*l
    A=torch.randn(M,K,device='cuda').half(); W=(torch.randn(3*C,K,device='cuda')*0.03).half(); b=torch.randn(3*C,device='cuda'); sm=torch.full((H,),1.4,device='cuda'); res=[]
    for r in range(6):
        qo=torch.empty(M,C,device='cuda',dtype=torch.float16); k2=torch.zeros(B2,H,Lmax,64,device='cuda',dtype=torch.float16); v2=torch.zeros_like(k2)
        hip.call('gemm_qkv_f16',A,K,W,K,b,M,C,K,sm,1.0,1,qo,k2,v2,B2,l,H,pos0,Lmax); res.append((qo.clone(),k2.clone(),v2.clone()))
    bad=[i for i in range(1,6) if not all(torch.equal(x,y) for x,y in zip(res[i],res[0]))]
    print(f'qkv16 l={l}: mismatching {bad}')
    x=torch.randn(M,C,device='cuda'); sc=torch.randn(B2,C,device='cuda'); sh=torch.randn(B2,C,device='cuda'); o=[]
    for r in range(4):
        y=torch.empty(M,C,device='cuda',dtype=torch.float16); hip.call('ln_modulate_f16out',x,sc,C,sh,C,y,M,C,l,1e-6); o.append(y.clone())
    print(f'ln16 l={l}: mismatching', [i for i in range(1,4) if not torch.equal(o[i],o[0])])
