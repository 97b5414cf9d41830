"""Context for the roofline fractions: what the vendor GEMM library (hipBLASLt/rocBLAS through torch.matmul, bf16 in, fp32
accumulate) reaches on the bare GEMM shapes of one launch of the chain kernels (262,144 points, F = 512), with nothing fused:
no sin/cos, no bias, no stash, no heads.  Measurement only - the product path never calls a library GEMM."""
import json
import torch

dev = torch.device("cuda:0")
M, F = 262144, 512
torch.manual_seed(0)
x = torch.randn(M, F, device=dev, dtype=torch.bfloat16)
g = torch.randn(M, F, device=dev, dtype=torch.bfloat16)
w = torch.randn(F, F, device=dev, dtype=torch.bfloat16) * 0.04
y = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
dw = torch.empty(F, F, device=dev, dtype=torch.bfloat16)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


flops = 2.0 * M * F * F
res = {}
ms = timed(lambda: torch.matmul(x, w.t(), out=y))
res["layer_forward  Y = X W^T   (262144x512 @ 512x512)"] = {"ms": ms, "tflops": flops / ms * 1e-9}
ms = timed(lambda: torch.matmul(g, w, out=y))
res["layer_backward dX = dZ W   (262144x512 @ 512x512)"] = {"ms": ms, "tflops": flops / ms * 1e-9}
ms = timed(lambda: torch.matmul(g.t(), x, out=dw))
res["weight_grad    dW = dZ^T X (512x262144 @ 262144x512)"] = {"ms": ms, "tflops": flops / ms * 1e-9}
print(json.dumps(res, indent=1))
