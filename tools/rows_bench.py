"""Development aid: the 64 -> 3 transposed layer of dcgan64 (g_h4 forward, d_h0 backward-data) per epilogue, same tensors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K, lib
d = torch.device("cuda:0")
B, H = int(os.environ.get("RB_B", "1024")), 32
x = torch.randn(B, H, H, 64, device=d); w = torch.randn(5, 5, 3, 64, device=d) * 0.02; b = torch.zeros(3, device=d)
y = torch.empty(B, 2 * H, 2 * H, 3, device=d); aux = torch.randn(B, 2 * H, 2 * H, 3, device=d).tanh()


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(2):
    for name, epi, kw in (("none", lib.EPI_NONE, {}), ("tanh", lib.EPI_TANH, {}), ("lrelu", lib.EPI_LRELU, {})):
        t = timeit(lambda: K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), 2, 2, epi, out=y))
        print(f"fwd {name:6s} {t:7.1f} us  {lib.last_kernel()}")
    for name, epi in (("none", lib.EPI_NONE), ("tanh'", lib.EPI_TANH_BWD), ("lrelu'", lib.EPI_LRELU_BWD)):
        t = timeit(lambda: K.conv2d_bwd_data(x, w, (2 * H, 2 * H), 2, 2, out=y, epilogue=epi, ep_aux=aux if epi != lib.EPI_NONE else None))
        print(f"bwd {name:6s} {t:7.1f} us  {lib.last_kernel()}")

# the 3 -> 64 forward-direction twin (d_h0 forward, g_h4 backward-data): conv_patch2_kernel
xi = torch.randn(B, 2 * H, 2 * H, 3, device=d).tanh(); wc = torch.randn(5, 5, 3, 64, device=d) * 0.02; bc = torch.zeros(64, device=d)
yo = torch.empty(B, H, H, 64, device=d); auxo = torch.randn(B, H, H, 64, device=d); a64 = torch.rand(64, device=d) + 0.5
for rep in range(2):
    for name, epi in (("none", lib.EPI_NONE), ("lrelu", lib.EPI_LRELU)):
        t = timeit(lambda: K.conv2d_fwd(xi, wc, bc, 2, 2, epi, out=yo))
        print(f"F fwd {name:8s} {t:7.1f} us  {lib.last_kernel()}")
    t = timeit(lambda: K.deconv2d_bwd_data(xi, wc, (H, H), 2, 2, out=yo, epilogue=lib.EPI_RELU_BWD_AFFINE, ep_a=a64, ep_aux=auxo))
    print(f"F bwd relu'*a   {t:7.1f} us  {lib.last_kernel()}")
