"""Fixed-batch training of the ResNet-50 bench configuration for many steps: the HIP engine (Gram form on / off by NKB_GRAM_BN)
next to the oracle module under torch.autocast(bfloat16) on the same GPU, same initial state and batch — loss every 25 steps.
A robustness check of the statistics-by-algebra path once the activations are far from their initial scale."""
import argparse, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
from oracle.torch_engine import make_optimizer
from oracle.torch_models import OracleClassifier
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
which = sys.argv[2] if len(sys.argv) > 2 else "hip"
args = argparse.Namespace(model="resnet50", classes=1000, batch=256, dtype="bf16", heads="")
dev = torch.device("cuda:0")
model, opt, crit = bench.build(args, dev)
OPT = dict(type="nadam", lr=1e-4, backbone_lr=1e-5, classifier_lr=1e-4, weight_decay=0.01, backbone_weight_decay=0.01, classifier_weight_decay=0.2)
g = torch.Generator().manual_seed(7)
img = torch.randn(256, 3, 224, 224, generator=g).to(dev); tgt = torch.randint(0, 1000, (256,), generator=g).to(dev)
if which == "torch":
    cfg = dict(task="single", model="resnet50", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0, classifier_initialization="kaiming_normal_")
    o = OracleClassifier(cfg, [str(i) for i in range(1000)]); o.load_state_dict(model.state_dict()); o = o.to(dev).train()
    del model, opt
    model, opt = o, make_optimizer(o, OPT)
    crit = lambda x, y: torch.nn.functional.cross_entropy(x.float(), y)
model.train()
out = []
for i in range(steps):
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(img), tgt)
    loss.backward(); opt.step()
    if i % 25 == 0 or i == steps - 1: out.append((i, loss.detach().clone()))
torch.cuda.synchronize()
print(which, os.environ.get("NKB_GRAM_BN", "1"), " ".join(f"{i}:{l.item():.3f}" for i, l in out), flush=True)
