import sys, os, torch
sys.path.insert(0, os.getcwd())
from multimodalsum_amd import kernels as kn
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dt = torch.bfloat16
for R, C in [(224*56*56, 64), (224*56*56, 256), (224*28*28, 512), (224*14*14, 256), (224*14*14, 1024)]:
    x = torch.randn(R, C, device="cuda").to(dt); res = torch.randn(R, C, device="cuda").to(dt); dy = torch.randn(R, C, device="cuda").to(dt)
    y = torch.empty_like(x); dx = torch.empty_like(x); dres = torch.empty_like(x)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    sums = torch.empty(2*C, device="cuda"); dsums = torch.empty(2*C, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    mb = R*C*2/1e6
    t1 = timeit(lambda: kn.bn_reduce(x, sums)); t2 = timeit(lambda: kn.bn_apply(x, sums, gamma, beta, res, y, rm, rv, 1e-5, 0.1, True, True))
    t3 = timeit(lambda: kn.bn_bwd_reduce(dy, y, x, sums, dsums, 1e-5, True)); t4 = timeit(lambda: kn.bn_bwd_apply(dy, y, x, sums, dsums, gamma, dx, dres, dg, db, 1e-5, True))
    print("R=%7d C=%4d (%5.0f MB)  reduce %6.1f us %4.2f TB/s | apply %6.1f us %4.2f TB/s | bwd_reduce %6.1f us %4.2f TB/s | bwd_apply %6.1f us %4.2f TB/s" % (R, C, mb, t1, mb/t1, t2, 3*mb/t2, t3, 3*mb/t3, t4, 5*mb/t4), flush=True)
