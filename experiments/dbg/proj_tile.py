import sys, torch
sys.path.insert(0, ".")
from tvts_amd import hip as K
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 192 * 785
for (N, Kd) in ((768, 768), (768, 3072)):
    As = [torch.randn(M, Kd, device="cuda").bfloat16() for _ in range(3)]
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    Rs = [torch.randn(M, N, device="cuda") for _ in range(3)]
    Os = [torch.empty(M, N, device="cuda") for _ in range(3)]
    for tile in (256, 128):
        i = [0]
        def f():
            i[0] = (i[0] + 1) % 3
            K.gemm_nt(As[i[0]], w, Os[i[0]], bias=bias, residual=Rs[i[0]], tile=tile)
        t = sorted(timeit(f) for _ in range(3))[1]
        gb = (M * Kd * 2 + M * N * 8) / 1e9
        print(f"fp32 out + fp32 residual {M}x{N}x{Kd} tile {tile}: {t:7.1f} us  {2.0 * M * N * Kd / t / 1e6:6.0f} TF  {gb / t * 1e3:5.2f} TB/s")
