"""Which attention path should the frozen white box use at fp32?  fwd+bwd time of one layer's attention at the
bench shape (B=64, H=12, S=617, d=64) for SDPA's backends vs an explicit baddbmm/softmax/bmm formulation."""
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

B, H, S, D = 64, 12, 617, 64
dev = "cuda"
q, k, v = (torch.randn(B, H, S, D, device=dev, requires_grad=True) for _ in range(3))
store = torch.zeros(1, H, S, 624, device=dev)
store[..., :S] = torch.randn(1, H, S, S, device=dev) * 0.02
bias = store[..., :S].expand(B, -1, -1, -1)
go = torch.randn(B, H, S, D, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run_sdpa(backend):
    def f():
        with sdpa_kernel(backend):
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias)
        o.backward(go)
        q.grad = k.grad = v.grad = None
    return f


def explicit():
    scale = D ** -0.5
    att = torch.baddbmm(bias.reshape(B * H, S, S) if False else bias.expand(B, H, S, S).reshape(B * H, S, S),
                        (q * scale).reshape(B * H, S, D), k.reshape(B * H, S, D).transpose(1, 2))
    p = torch.softmax(att, dim=-1)
    o = torch.bmm(p, v.reshape(B * H, S, D)).reshape(B, H, S, D)
    o.backward(go)
    q.grad = k.grad = v.grad = None


for name, be in (("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH), ("flash", SDPBackend.FLASH_ATTENTION)):
    try:
        print(name, "%.2f ms fwd+bwd" % timeit(run_sdpa(be)), flush=True)
    except Exception as e:
        print(name, "unavailable:", str(e)[:120], flush=True)
print("explicit baddbmm/softmax/bmm %.2f ms fwd+bwd" % timeit(explicit), flush=True)
