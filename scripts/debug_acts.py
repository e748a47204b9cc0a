import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification.model import get_model
from nkb_classification.losses import get_loss
from oracle.torch_models import OracleClassifier
backbone = sys.argv[1]; hw = int(sys.argv[2])
cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0, classifier_initialization="kaiming_normal_", task="single")
torch.manual_seed(0)
o32 = OracleClassifier(cfg_model, ["a","b","c"])
model = get_model(cfg_model, ["a","b","c"], "cuda:0")
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for p in o32.parameters():
        if p.dim() == 1: p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.5)
model.load_state_dict(o32.state_dict())
o64 = OracleClassifier(cfg_model, ["a","b","c"]).double(); o64.load_state_dict(o32.state_dict())
x = torch.randn(4, 3, hw, hw, generator=g); y = torch.randint(0, 3, (4,), generator=g)
acts = {}
def hook(name):
    def f(m, i, o): acts[name] = o.detach()
    return f
em = o64.emb_model
em.act1.register_forward_hook(hook("stem"))
for li in range(1, 5):
    for bi, blk in enumerate(getattr(em, f"layer{li}")):
        nst = 3 if hasattr(blk, "conv3") else 2
        for k in range(1, nst + 1):
            getattr(blk, f"act{k}").register_forward_hook(hook(f"layer{li}.{bi}.{k-1}"))
acts32 = {}
def hook32(name):
    def f(m, i, o): acts32[name] = o.detach()
    return f
em32 = o32.emb_model
em32.act1.register_forward_hook(hook32("stem"))
for li in range(1, 5):
    for bi, blk in enumerate(getattr(em32, f"layer{li}")):
        nst = 3 if hasattr(blk, "conv3") else 2
        for k in range(1, nst + 1):
            getattr(blk, f"act{k}").register_forward_hook(hook32(f"layer{li}.{bi}.{k-1}"))
o64.train(); model.train(); o32.train()
o64(x.double()); o32(x)
model(x.cuda()); torch.cuda.synchronize()
eng = model._active
for name, ref in acts.items():
    yh = eng.saved[name]["y"].float().cpu().permute(0, 3, 1, 2).double()
    y32 = acts32[name].double()
    fl_h = ((yh > 0) != (ref > 0)).sum().item(); fl_c = ((y32 > 0) != (ref > 0)).sum().item()
    print(f"{name:14s} n={ref.numel():8d} maxdiff hip {((yh-ref).abs().max()/ref.abs().max()).item():.2e} cpu32 {((y32-ref).abs().max()/ref.abs().max()).item():.2e} flips hip {fl_h} cpu {fl_c}  zeros {int((ref==0).sum())}")
