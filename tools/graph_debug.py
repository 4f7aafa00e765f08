import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyct_dram_emph_subtype_amd import med3d
from bodyct_dram_emph_subtype_amd.graph import GraphedTrainStep
from bodyct_dram_emph_subtype_amd.models import cls_train_loss
from bodyct_dram_emph_subtype_amd.optim import FusedAdam
DEV = "cuda:0"
g = torch.Generator().manual_seed(3)
batches = [(torch.randn(2, 1, 16, 32, 32, generator=g).to(DEV), (torch.rand(2, 1, 16, 32, 32, generator=g) > 0.3).float().to(DEV),
            torch.randint(0, 6, (2,), generator=g).to(DEV), torch.randint(0, 3, (2,), generator=g).to(DEV)) for _ in range(4)]
cw, pw = torch.full((6,), 1 / 6, device=DEV), torch.full((3,), 1 / 3, device=DEV)
def run(mode):
    torch.manual_seed(11)
    m = med3d.resnet18segcls(n_classes=[6, 3]).to(DEV).train()
    opt = FusedAdam(m.parameters(), lr=1e-3, capturable=(mode != "plain"))
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5) if os.environ.get("SCHED") else None
    print(mode, "lr after sched ctor", opt.param_groups[0]["lr"], {k: v for k, v in opt.param_groups[0].items() if k != "params"})
    def loss_fn(image, lung, cle, pse):
        return cls_train_loss(m(image, lung)[1], cle, pse, cw, pw)[0]
    def eager(b):
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(*b); loss.backward(); opt.step()
        return loss.detach().clone()
    out = []
    if mode == "graph":
        step = GraphedTrainStep(m, opt, loss_fn, batches[0], warmup=2)
        print("after capture: conv1 w sum", float(m.conv1.weight.double().sum()), "nbt", int(m.bn1.num_batches_tracked), "hyper", opt._hyper.tolist())
    else:
        out.append(float(eager(batches[0]))); out.append(float(eager(batches[0])))
        print(mode, "after 2 eager: conv1 w sum", float(m.conv1.weight.double().sum()), "nbt", int(m.bn1.num_batches_tracked), "hyper", None if opt._hyper is None else opt._hyper.tolist())
        step = lambda *b: eager(b)
    for b in batches[1:]:
        out.append(float(step(*b)))
    print(mode, out, float(m.conv1.weight.double().sum()))
for mode in ("graph", "eager-capturable", "plain"):
    run(mode)
