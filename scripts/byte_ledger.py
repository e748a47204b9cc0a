"""Per-tensor byte ledger of one ResNet-50 train step as THIS engine executes it: for every stage, which kernel
writes and which kernels read each of c (raw conv output), y (activation), g / dy' (masked gradients), dc (BN-backward output), the
mask bits and the Gram-form operands — algorithmic bytes, each tensor pass counted once per kernel that makes it.  Summed per kernel
family it is the floor the PMC counters (scripts/pmc_traffic.py) can be held against; the difference is over-fetch (operand panels
re-read across channel tiles, halos, write-allocate) plus the small tensors not modelled here (slabs, statistics, weights).

Usage: python scripts/byte_ledger.py [--batch 256] [--gram-max-c 128] [--pmc profiles/r03_resnet50_bf16_pmc_traffic.json]
  --gram-max-c 0 prints the round-2 schedule (no Gram-form closing stages) for comparison."""
import argparse
import collections
import json

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--gram-max-c", type=int, default=128)
ap.add_argument("--pmc", default=None)
args = ap.parse_args()
B, S = args.batch, 2            # bf16

led = collections.defaultdict(float)      # family -> bytes
rows = []                                 # (stage, kernel, family, reads, writes)


def add(stage, kernel, fam, reads, writes):
    led[fam] += reads + writes
    rows.append((stage, kernel, fam, reads, writes))


def T(hw, c):                             # bytes of a [B, hw, hw, c] bf16 tensor
    return B * hw * hw * c * S


def bottleneck(name, hw_in, cin, w, stride, first, gram, last_block):
    hw = hw_in // stride
    x, c1, c2, c3 = T(hw_in, cin), T(hw_in, w), T(hw, w), T(hw, 4 * w)
    bits = c3 / 16
    # ---------------- forward
    add(name, "conv1 1x1 (+stats)", "conv_fwd", x, c1)
    add(name, "bn_apply1", "bn_apply", c1, c1)
    add(name, f"conv2 3x3/s{stride} (+stats)", "conv_fwd", c1, c2)
    add(name, "bn_apply2" + (" + Gram" if gram else ""), "bn_apply", c2, c2)
    if first:
        xs = T(hw, cin)
        add(name, "downsample conv (+stats)", "conv_fwd", xs, c3)
    if gram:
        add(name, "conv3 + bn3 + shortcut + ReLU + bits (Gram form)", "conv_fwd", c2 + c3, c3 + bits)   # reads a and the shortcut
    else:
        add(name, "conv3 1x1 (+stats)", "conv_fwd", c2, c3)
        add(name, "bn_apply3 + shortcut + ReLU + bits", "bn_apply", 2 * c3, c3 + bits)
    # ---------------- backward (g = masked gradient of the block output, produced by the NEXT block's conv1 dgrad epilogue)
    if last_block:
        add(name, "bn_backward3 (reduce + apply, from avg-pool gradient)", "bn_bwd", 2 * (c3 + c3), c3)
        add(name, "conv3 dgrad (+ mask, bn2 sums)", "conv_dgrad", c3 + c2, c2)
        add(name, "conv3 wgrad", "conv_wgrad", c3 + c2, 0)
    elif gram:
        add(name, "R = g^T a (weight-gradient GEMM, main stream)", "conv_wgrad", c3 + c2, 0)
        add(name, "conv3 dgrad over [g | a] (+ mask, bn2 sums)", "conv_dgrad", c3 + c2 + c2, c2)
    else:
        add(name, "bn_bwd_apply3", "bn_bwd", c3 + c3, c3)
        add(name, "conv3 dgrad (+ mask, bn2 sums)", "conv_dgrad", c3 + c2, c2)
        add(name, "conv3 wgrad", "conv_wgrad", c3 + c2, 0)
    add(name, "bn_bwd_apply2", "bn_bwd", c2 + c2, c2)
    add(name, f"conv2 dgrad 3x3/s{stride} (+ mask, bn1 sums)", "conv_dgrad", c2 + c1, c1)
    add(name, "conv2 wgrad", "conv_wgrad", c2 + c1, 0)
    add(name, "bn_bwd_apply1", "bn_bwd", c1 + c1, c1)
    if first:
        add(name, "downsample bn_backward (reduce + apply)", "bn_bwd", 2 * (c3 + c3), c3)
        add(name, "downsample dgrad (sub-grid GEMM)", "conv_dgrad", c3, T(hw, cin))
        add(name, "downsample wgrad", "conv_wgrad", c3 + T(hw, cin), 0)
    # conv1 dgrad: produces the PREVIOUS block's g (adds the shortcut gradient, applies that block's bits; non-Gram predecessor:
    # also reads its c3 for the BN sums)
    add(name, "conv1 dgrad + shortcut gradient + previous block's mask (+ sums)", "conv_dgrad", c1 + x + x / 16, x)
    add(name, "conv1 wgrad", "conv_wgrad", c1 + x, 0)
    return hw


# stem: packed image -> conv 7x7/2 -> bn -> relu -> maxpool
img, xp = B * 3 * 224 * 224 * 4, B * 224 * 224 * 4 * S
c0, p0 = T(112, 64), T(56, 64)
add("stem", "stem_pack", "stem", img, xp)
add("stem", "stem conv 7x7/2 (+stats)", "conv_fwd", xp, c0)
add("stem", "bn -> relu -> maxpool", "stem_tail", c0, p0 + p0 / 2)
add("stem", "maxpool/relu/bn backward reduce", "stem_tail", p0 + p0 / 2 + c0, 0)
add("stem", "maxpool/relu/bn backward apply", "stem_tail", p0 + p0 / 2 + c0, c0)
add("stem", "stem wgrad", "conv_wgrad", c0 + xp, 0)

hw, cin = 56, 64
layers = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]
nblk = sum(n for _, n, _ in layers)
k = 0
prev_gram = []
for li, (w, n, stride) in enumerate(layers):
    for b in range(n):
        k += 1
        gram = w <= args.gram_max_c and k < nblk
        hw_in = hw
        hw = bottleneck(f"layer{li + 1}.{b}", hw, cin, w, stride if b == 0 else 1, b == 0, gram, k == nblk)
        if not gram and k < nblk:
            # the next block's conv1 dgrad epilogue reads this block's c3 for the BN sums (non-Gram closing stage)
            add(f"layer{li + 1}.{b}", "(+ c3 read by the next block's conv1 dgrad epilogue)", "conv_dgrad", T(hw, 4 * w), 0)
        cin = 4 * w
params = 25.56e6
add("optimizer", "NAdam over the arena (+ bf16 shadow)", "optim", params * 22, params * 14)

total = sum(led.values())
print(f"ResNet-50, batch {B}, bf16, Gram form up to {args.gram_max_c} channels: algorithmic bytes per step {total / 1e9:.2f} GB")
for fam, v in sorted(led.items(), key=lambda kv: -kv[1]):
    print(f"  {fam:12s} {v / 1e9:7.2f} GB")
if args.pmc:
    doc = json.load(open(args.pmc))
    ks = doc["kernels"]
    fam_map = {"conv_fwd": ("conv_igemm_fwd", "gemm8p_fwd"), "conv_dgrad": ("conv_igemm_bwd", "gemm8p_bwd"),
               "conv_wgrad": ("conv_wgrad", "wgrad8p", "wgrad3x3", "wgradr"), "bn_apply": ("bn_apply",), "bn_bwd": ("bn_bwd_apply", "bn_bwd_reduce"),
               "stem_tail": ("stem_tail",), "stem": ("stem",), "optim": ("optim",)}
    print(f"\nagainst the PMC passes ({args.pmc}): measured {doc['step_total_bytes'] / 1e9:.2f} GB per step")
    print(f"  {'family':12s} {'ledger GB':>10s} {'measured GB':>12s} {'ratio':>6s}")
    for fam, keys in fam_map.items():
        m = sum(ks[q]["bytes_per_step"] for q in keys if q in ks)
        print(f"  {fam:12s} {led[fam] / 1e9:10.2f} {m / 1e9:12.2f} {m / max(led[fam], 1):6.2f}")
print("\nper stage (GB read / written), first block of each layer:")
for st, kern, fam, r, wr in rows:
    if st.endswith(".0") or st in ("stem", "optimizer"):
        print(f"  {st:10s} {kern:70s} {r / 1e9:6.3f} {wr / 1e9:6.3f}")
