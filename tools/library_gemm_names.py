"""Which library kernels does torch.matmul pick for the headline's fp32 shapes?  (run under rocprofv3 --kernel-trace; names carry the macro tile)"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for M, N, K in ((25216, 1152, 384), (12672, 1152, 384), (25216, 384, 384), (25216, 1536, 384), (25216, 384, 1536), (36928, 768, 3072), (36928, 3072, 768)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    for _ in range(3): torch.matmul(x, w.t(), out=out)
torch.cuda.synchronize()
