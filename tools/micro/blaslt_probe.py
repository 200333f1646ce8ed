"""Which hipBLASLt kernels torch.mm picks for the step's weight-gradient shapes (run under
rocprofv3 --kernel-trace --stats)."""
import torch
dev = "cuda"
for (N, K, B) in ((1000, 1000, 4096), (1000, 368, 4096), (368, 368, 4096)):
    dy, x = torch.randn(B, N, device=dev), torch.randn(B, K, device=dev)
    out = torch.empty(N, K, device=dev)
    for _ in range(20):
        torch.mm(dy.t(), x, out=out)
torch.cuda.synchronize()
