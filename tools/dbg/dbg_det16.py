import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from var_amd import hip
torch.manual_seed(0)
def conv(B,H,W,Cin,Cout,res,omode=0,reps=6):
    x=torch.randn(B,H,W,Cin,device='cuda').half(); w=(torch.randn(Cout,3,3,Cin,device='cuda')*0.02).half(); b=torch.randn(Cout,device='cuda')
    r=torch.randn(B,H,W,Cout,device='cuda').half() if res else None
    nblk=hip.conv_gn_blocks(H,W,Cout) if (omode==0 and Cout%4==0) else 0
    outs=[]
    for i in range(reps):
        part=torch.zeros(B,nblk,Cout,2,dtype=torch.float64,device='cuda') if nblk else None
        out=torch.empty((B,Cout,H,W),dtype=torch.float32,device='cuda') if omode else torch.empty(B,H,W,Cout,dtype=torch.float16,device='cuda')
        hip.call('conv3x3_nhwc_f16',x,w,b,r,out,part,B,H,W,Cin,Cout,omode)
        outs.append((out.clone(), None if part is None else part.clone()))
    torch.cuda.synchronize()
    bad=[i for i in range(1,reps) if not torch.equal(outs[i][0],outs[0][0]) or (nblk and not torch.equal(outs[i][1],outs[0][1]))]
    nd = int((outs[bad[0]][0]!=outs[0][0]).sum()) if bad else 0
    print(f'conv16 B{B} {H}x{W} {Cin}->{Cout} res{res} omode{omode}: mismatching reps {bad} ndiff {nd}')
def up(B,H,W,Cin,Cout,reps=6):
    x=torch.randn(B,H//2,W//2,Cin,device='cuda').half(); wp=(torch.randn(4,Cout,2,2,Cin,device='cuda')*0.02).half(); b=torch.randn(Cout,device='cuda')
    nblk=hip.conv_gn_blocks(H,W,Cout,phase=True); outs=[]
    for i in range(reps):
        part=torch.zeros(B,nblk,Cout,2,dtype=torch.float64,device='cuda'); out=torch.empty(B,H,W,Cout,dtype=torch.float16,device='cuda')
        hip.call('upconv_phase_f16',x,wp,b,out,part,B,H,W,Cin,Cout); outs.append((out.clone(),part.clone()))
    bad=[i for i in range(1,reps) if not torch.equal(outs[i][0],outs[0][0]) or not torch.equal(outs[i][1],outs[0][1])]
    print(f'upconv16 B{B} {H}x{W} {Cin}->{Cout}: mismatching reps {bad}')
for B in (2,4):
    conv(B,16,16,32,32,0); conv(B,16,16,32,640,0); conv(B,16,16,640,640,1); conv(B,32,32,640,320,0); conv(B,64,64,320,320,1); conv(B,128,128,320,160,0)
    conv(B,128,128,160,160,1); conv(B,256,256,160,160,1); conv(B,256,256,160,3,0,1)
    up(B,32,32,640,640); up(B,64,64,320,320); up(B,128,128,320,320); up(B,256,256,160,160)
def gemm(M,N,K,epi,reps=5):
    A=torch.randn(M,K,device='cuda').half(); W=(torch.randn(N,K,device='cuda')*0.03).half(); b=torch.randn(N,device='cuda'); r=torch.randn(M,N,device='cuda'); outs=[]
    for i in range(reps):
        out=torch.empty(M,N,device='cuda',dtype=torch.float32 if epi==2 else torch.float16)
        hip.call('gemm_nt_f16',A,K,W,K,b,out,N,0 if epi==2 else 1,M,N,K,epi,r if epi==2 else None,N,0,None,0,1,1,0,0,0); outs.append(out.clone())
    print(f'gemm16 {M}x{N}x{K} epi{epi}: mismatching', [i for i in range(1,reps) if not torch.equal(outs[i],outs[0])])
gemm(2048,3072,1024,0); gemm(2048,1024,4096,2); gemm(128,4096,1024,1); gemm(800,1024,1024,2)
