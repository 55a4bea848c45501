import sys, numpy as np, torch
sys.path.insert(0, '.')
from adm_amd import hip
from adm_amd.hip import call, ptr
B,H,W,C=2,4,4,32
torch.manual_seed(0)
x=torch.randn(B,H,W,C,device='cuda'); dy=torch.randn(B,H,W,C,device='cuda')
wx=torch.zeros(C,12,C,device='cuda')
call("adm_conv_wgrad_wino2d", ptr(x), ptr(dy), ptr(wx), None, B,H,W,C,C,C,C, 1)
torch.cuda.synchronize()
xn=x.cpu().numpy().astype(np.float64); dn=dy.cpu().numpy().astype(np.float64)
xp=np.zeros((B,H+2,W+2,C)); xp[:,1:-1,1:-1]=xn
A=np.array([[1,0],[1,1],[1,-1],[0,-1.]]); Bt=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1.]]); Gt=np.array([[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1.]])
m=np.zeros((4,4,C,C))
for b in range(B):
  for ty in range(H//2):
    for tx in range(W//2):
      e=dn[b,2*ty:2*ty+2,2*tx:2*tx+2]          # [2,2,co]
      d=xp[b,2*ty:2*ty+4,2*tx:2*tx+4]          # [4,4,ci]
      a=np.einsum('ip,pqc,jq->ijc',A,e,A)
      bb=np.einsum('ip,pqc,jq->ijc',Bt,d,Bt)
      m+=np.einsum('ijc,ijd->ijcd',a,bb)
want=np.einsum('kx,yxcd->ykcd',Gt,m)          # x-fold: [ey][kx][co][ci]
got=wx.cpu().numpy().reshape(C,4,3,C).transpose(1,2,0,3)
for ey in range(4):
    print(ey, np.abs(got[ey]-want[ey]).max(), np.abs(want[ey]).max())
