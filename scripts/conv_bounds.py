"""Analytic lower bounds for the ResNet-50 (bs 256, bf16) convolutions: HBM bytes vs MFMA flops per pass."""
B = 256
HBM = 5.0e12      # achievable streaming rate
MFMA = 2.5e15     # dense bf16 peak
convs = []        # (name, M_out, Cin, Cout, R, stride, M_in)
def add(name, hin, cin, cout, r, stride):
    hout = hin // stride
    convs.append((name, B * hout * hout, cin, cout, r, stride, B * hin * hin))
    return hout
h = 56
inpl = 64
for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
    for b in range(blocks):
        s = stride if b == 0 else 1
        add(f"l{li+1}.{b}.c1", h, inpl, planes, 1, 1)
        h2 = add(f"l{li+1}.{b}.c2", h, planes, planes, 3, s)
        add(f"l{li+1}.{b}.c3", h2, planes, planes * 4, 1, 1)
        if b == 0:
            add(f"l{li+1}.{b}.ds", h, inpl, planes * 4, 1, s)
        inpl = planes * 4
        h = h2
tot = dict(fwd=[0, 0, 0], dgrad=[0, 0, 0], wgrad=[0, 0, 0])
for name, mo, ci, co, r, s, mi in convs:
    fl = 2.0 * mo * ci * co * r * r
    x, y, w = mi * ci * 2, mo * co * 2, ci * co * r * r * 2
    for kind, byts in (("fwd", x + y + w), ("dgrad", x + y + w), ("wgrad", x + y + w * 2)):
        t_mem, t_mfma = byts / HBM, fl / MFMA
        tot[kind][0] += t_mem; tot[kind][1] += t_mfma; tot[kind][2] += max(t_mem, t_mfma)
for k, (tm, tf, tb) in tot.items():
    print(f"{k:6s} sum(hbm-bound)={tm*1e3:6.2f} ms  sum(mfma-bound)={tf*1e3:6.2f} ms  sum(max)={tb*1e3:6.2f} ms")
