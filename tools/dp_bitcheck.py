"""GPU box: is the data-parallel train step bit-identical to the plain one, step by step?
   python tools/dp_bitcheck.py
Runs five optimizer steps of resnet18segreg on a 16x32x32 batch four ways (plain twice, data parallel at world size 1 with
every collective forced twice; FusedAdam capturable and not) and prints, per step, how many gradient tensors and
state_dict entries differ between the runs.  Found the Adam float4 / scalar path discrepancy of round 5:
gradients equal, updated BatchNorm parameters one ulp apart."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="35123"; os.environ["DRAM_TUNING"]="1"
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from bodyct_dram_emph_subtype_amd import distributed as ddist, med3d
from bodyct_dram_emph_subtype_amd.optim import FusedAdam
g = torch.Generator().manual_seed(501)
x = torch.randn(2,1,16,32,32, generator=g).cuda(); lungs=(torch.rand(2,1,16,32,32, generator=g)>0.3).float().cuda()
def loss_of(dense, outs): return outs[0].sum() - 0.5*outs[1].sum() + 0.1*(dense[0]*dense[1]).mean()
def run(mode, capturable=True, nsteps=5):
    torch.manual_seed(4)
    m = med3d.resnet18segreg().cuda().train()
    ddist.attach(m, bucket_bytes=8<<20, force=(mode!="plain"))
    opt = FusedAdam(m.parameters(), lr=1e-3, capturable=capturable)
    hist=[]
    for i in range(nsteps):
        opt.zero_grad(set_to_none=True)
        d,o = m(x,lungs); l = loss_of(d,o); l.backward()
        torch.cuda.synchronize()
        grads={n:p.grad.detach().clone() for n,p in m.named_parameters()}
        opt.step(); torch.cuda.synchronize()
        hist.append((float(l), grads, {k:v.detach().clone() for k,v in m.state_dict().items()}))
    return hist
a=run("plain"); b=run("plain"); c=run("dp"); d=run("dp")
def cmp(h1,h2,tag):
    for i,(s1,s2) in enumerate(zip(h1,h2)):
        bad_g=[n for n in s1[1] if not torch.equal(s1[1][n],s2[1][n])]
        bad_s=[k for k in s1[2] if not torch.equal(s1[2][k],s2[2][k])]
        print(tag,"step",i,"loss",s1[0]==s2[0],"grads differing",len(bad_g),bad_g[:4],"state differing",len(bad_s),bad_s[:4])
cmp(a,b,"plain-vs-plain"); cmp(c,d,"dp-vs-dp"); cmp(a,c,"plain-vs-dp")
e=run("plain",False); f=run("dp",False); cmp(e,f,"noncapturable plain-vs-dp")
dist.destroy_process_group()
