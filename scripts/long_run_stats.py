"""After N fixed-batch steps of the ResNet-50 bench configuration on the HIP engine and, from the same initial state, on the oracle
under torch.autocast: (1) BatchNorm running statistics of the two, per layer (relative L2 difference), (2) eval-mode logits of the
HIP model against the oracle LOADED WITH THE HIP MODEL'S state_dict (the folded-BatchNorm eval path on trained statistics)."""
import argparse, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
from oracle.torch_engine import make_optimizer
from oracle.torch_models import OracleClassifier
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = argparse.Namespace(model="resnet50", classes=1000, batch=256, dtype="bf16", heads="")
dev = torch.device("cuda:0")
model, opt, crit = bench.build(args, dev)
OPT = dict(type="nadam", lr=1e-4, backbone_lr=1e-5, classifier_lr=1e-4, weight_decay=0.01, backbone_weight_decay=0.01, classifier_weight_decay=0.2)
cfg = dict(task="single", model="resnet50", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0, classifier_initialization="kaiming_normal_")
o = OracleClassifier(cfg, [str(i) for i in range(1000)]); o.load_state_dict(model.state_dict()); o = o.to(dev).train()
oo = make_optimizer(o, OPT)
g = torch.Generator().manual_seed(7)
img = torch.randn(256, 3, 224, 224, generator=g).to(dev); tgt = torch.randint(0, 1000, (256,), generator=g).to(dev)
model.train()
for i in range(steps):
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(img), tgt)
    loss.backward(); opt.step()
    oo.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lo = torch.nn.functional.cross_entropy(o(img).float(), tgt)
    lo.backward(); oo.step()
torch.cuda.synchronize()
print(f"after {steps} steps: loss hip {loss.item():.4f} torch {lo.item():.4f}")
hb, ob = dict(model.named_buffers()), dict(o.named_buffers())
worst = []
for k in hb:
    if "running" in k:
        d = ((hb[k].float() - ob[k].float()).norm() / ob[k].float().norm().clamp_min(1e-12)).item()
        worst.append((d, k))
worst.sort(reverse=True)
print("running statistics, relative L2 hip vs torch, five worst:", [(round(d, 4), k) for d, k in worst[:5]], "median", round(sorted(d for d, _ in worst)[len(worst) // 2], 5))
o2 = OracleClassifier(cfg, [str(i) for i in range(1000)]); o2.load_state_dict(model.state_dict()); o2 = o2.to(dev).eval()
model.eval()
xe = torch.randn(64, 3, 224, 224, generator=g).to(dev)
with torch.no_grad():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = model(xe).float()
    b = o2(xe).float()            # fp32 eval of the same weights / statistics
rel = ((a - b).abs().max() / b.abs().max()).item()
print(f"eval logits (bf16 HIP vs fp32 torch, same state_dict): rel max err {rel:.3e}, argmax agreement {(a.argmax(1) == b.argmax(1)).float().mean().item():.3f}")
