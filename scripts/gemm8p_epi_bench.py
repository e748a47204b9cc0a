"""The eight GEMM launches of one ViT-B/16 block (forward + data gradients, batch 256) with the epilogues the train step gives them,
run as a chain over block-sized buffers (> 256 MB per pass: nothing stays in the Infinity Cache between passes), per-launch device
times from the library's HIP-event profiler.  `python scripts/gemm8p_epi_bench.py [reps]`; environment switches are inherited, so
alternating builds / settings in one process pool is `scripts/ab_lib.sh`'s job."""
import os, sys, statistics, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
M, D, H3, H4 = 256 * 197, 768, 2304, 3072
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(T)
x, res, o = rn(M, D), rn(M, D), rn(M, D)
w_qkv, w_proj, w_fc1, w_fc2 = rn(H3, D) * 0.03, rn(D, D) * 0.03, rn(H4, D) * 0.03, rn(D, H4) * 0.02
wd_qkv, wd_proj, wd_fc1, wd_fc2 = w_qkv.t().contiguous(), w_proj.t().contiguous(), w_fc1.t().contiguous(), w_fc2.t().contiguous()
b3, b1, b4 = torch.zeros(H3, device=dev), torch.zeros(D, device=dev), torch.zeros(H4, device=dev)
qkv, y1, u, du, y2 = torch.empty(M, H3, device=dev, dtype=T), torch.empty(M, D, device=dev, dtype=T), torch.empty(M, H4, device=dev, dtype=T), torch.empty(M, H4, device=dev, dtype=T), torch.empty(M, D, device=dev, dtype=T)
gq, dpre, dh, do_ = rn(M, H3), torch.empty(M, H4, device=dev, dtype=T), torch.empty(M, D, device=dev, dtype=T), torch.empty(M, D, device=dev, dtype=T)
gemm = lambda xx, ww, yy, K, N, **kw: hip.conv_gemm(d, 0, xx, ww, yy, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N, **kw)
OPS = [
    ("qkv fwd  768->2304 +bias      ", 2.0 * M * D * H3, lambda: gemm(x, w_qkv, qkv, D, H3, bias=b3)),
    ("proj fwd 768->768  +bias +res ", 2.0 * M * D * D, lambda: gemm(o, w_proj, y1, D, D, bias=b1, add=res, ldadd=D)),
    ("fc1 fwd  768->3072 gelu, gelu'", 2.0 * M * D * H4, lambda: hip.linear_gelu(d, 5, y1, w_fc1, b4, None, u, du, M, D, H4)),
    ("fc2 fwd  3072->768 +bias +res ", 2.0 * M * D * H4, lambda: gemm(u, w_fc2, y2, H4, D, bias=b1, add=y1, ldadd=D)),
    ("fc2 dgrad 768->3072 * gelu'   ", 2.0 * M * D * H4, lambda: hip.linear_gelu(d, 4, y2, wd_fc2, None, du, dpre, None, M, D, H4)),
    ("fc1 dgrad 3072->768           ", 2.0 * M * D * H4, lambda: gemm(dpre, wd_fc1, dh, H4, D)),
    ("proj dgrad 768->768           ", 2.0 * M * D * D, lambda: gemm(dh, wd_proj, do_, D, D)),
    ("qkv dgrad 2304->768           ", 2.0 * M * D * H3, lambda: gemm(gq, wd_qkv, dh, H3, D)),
]
for _, _, f in OPS: f()
torch.cuda.synchronize()
for _ in range(2):
    for _, _, f in OPS: f()
torch.cuda.synchronize()
hip.prof_enable(True)
for _ in range(reps):
    for _, _, f in OPS: f()
torch.cuda.synchronize(); hip.prof_enable(False)
raw = hip.prof_collect_raw()
assert len(raw) == reps * len(OPS), (len(raw), reps)
tot = 0.0
for i, (name, fl, _) in enumerate(OPS):
    ts = [raw[r * len(OPS) + i][1] * 1e3 for r in range(reps)]
    med = statistics.median(ts); tot += med
    print(f"{name} {med:7.1f} us (min {min(ts):6.1f})  {fl / med / 1e6:7.1f} TF/s", flush=True)
print(f"block GEMM chain: {tot:7.1f} us  ({tot * 12 / 1e3:.2f} ms per 12-block step)  tag={os.environ.get('TAG', '')}")
