"""which hipBLASLt solutions torch.matmul picks for the yardstick shapes (kernel names carry macro tile / unroll / wave layout).
Study only: nothing of hipBLASLt is linked or called by the product."""
import torch
dev = torch.device("cuda:0")
for (m, n, k) in [(8192, 8192, 8192), (4096, 4096, 4096), (20736, 1024, 1024), (20736, 3072, 1024), (20736, 4096, 1024), (20736, 1024, 4096)]:
    a = (torch.rand(m, k, device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, k, device=dev) * 2 - 1).to(torch.bfloat16)
    for _ in range(3):
        c = torch.matmul(a, b.t())
    torch.cuda.synchronize()
