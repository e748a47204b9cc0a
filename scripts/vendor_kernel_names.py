"""Which hipBLASLt kernels does torch.mm pick on the ViT / ResNet GEMM shapes (run under rocprofv3 --kernel-trace)?"""
import torch
for (M, K, N) in [(50432, 768, 768), (50432, 768, 2304), (50432, 3072, 768), (50176, 1024, 256), (12544, 2048, 512)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(x, w.t(), out=y)
torch.cuda.synchronize()
