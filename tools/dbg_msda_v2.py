import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd._lib import lib, ptr, cur_stream, check
torch.manual_seed(0)
shapes=[(12,20),(6,10),(3,5),(2,3)]; starts=[0,240,300,315]; N=321
B,M,D,L,P=1,8,32,4,4
lv=([s[0] for s in shapes],[s[1] for s in shapes],starts)
def run(proj, ref, value_rows):
    n=4; arr=lambda v:(ctypes.c_int*n)(*[int(x) for x in v])
    out=torch.empty(B*N,256,device="cuda")
    check(lib.mdqe_msda_fused_f32(ptr(proj),640,N,None,ptr(proj[:,256:]),640,ptr(proj[:,512:]),640,ptr(ref),0,2,0,None,arr(lv[0]),arr(lv[1]),arr(lv[2]),B,M,D,1,L,N,P,1.0,ptr(out),256,value_rows,cur_stream()),"x")
    return out
ref=torch.full((N,2),0.5,device="cuda")
# case 1: value=1, offsets 0, logits 0 -> every sample at the centre: out = 1
proj=torch.zeros(B*N,640,device="cuda"); proj[:,:256]=1
a=run(proj,ref,0); b=run(proj,ref,B*N); print("case1 v1", a[0,:4].tolist(), "v2", b[0,:4].tolist())
# case 2: value = channel index
proj[:,:256]=torch.arange(256,device="cuda").float()[None]
a=run(proj,ref,0); b=run(proj,ref,B*N); print("case2 v1", a[0,[0,1,33,255]].tolist(), "v2", b[0,[0,1,33,255]].tolist())
# case 3: value = row index (pixel id), centre sample
proj[:,:256]=torch.arange(N,device="cuda").float()[:,None]
a=run(proj,ref,0); b=run(proj,ref,B*N); print("case3 v1", a[0,:2].tolist(), "v2", b[0,:2].tolist())
# case 4: logits random
proj[:,512:]=torch.randn(N,128,device="cuda")
a=run(proj,ref,0); b=run(proj,ref,B*N); print("case4 v1", a[0,:2].tolist(), a[5,40:42].tolist(), "v2", b[0,:2].tolist(), b[5,40:42].tolist())
# case 5: offsets random small
proj[:,256:512]=torch.randn(N,256,device="cuda")*0.3
a=run(proj,ref,0); b=run(proj,ref,B*N); print("case5 maxdiff", (a-b).abs().max().item(), a[0,:2].tolist(), b[0,:2].tolist())
