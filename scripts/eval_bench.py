"""Eval-mode forward throughput (val_epoch's per-batch body: forward + loss under no_grad), ResNet-50 bs 256 bf16.
NKB_EVAL_FOLD=0/1 switches the folded-BatchNorm fast path."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification.model import get_model
from nkb_classification.losses import get_loss
name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
model = get_model(dict(task="single", model=name, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                       classifier_initialization="kaiming_normal_"), [str(i) for i in range(1000)], dev).eval()
crit = get_loss(dict(task="single", type="CrossEntropyLoss"), dev)
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (B,), device=dev)
def step():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return crit(model(x), y)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 30
for _ in range(n): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{name} eval bs{B} bf16 fold={os.environ.get('NKB_EVAL_FOLD', '1')}: {1e3 * dt / n:.2f} ms/batch, {B * n / dt:.0f} img/s")
