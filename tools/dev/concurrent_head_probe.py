"""Would the SSIM forward and the per-pixel pass overlap if they ran on two streams?  (They are independent up to the side jobs
of the per-pixel pass's first two workgroups.)  Times hgs_ssim_l1_forward and hgs_orientation_loss_forward on 1080p planes back to
back on one stream, and concurrently on two, eagerly and as a captured graph with a fork / join."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import hgs_runtime as rt
from hgs_runtime import fused
rt.lib()
dev = torch.device("cuda")
H, W = 1080, 1920
torch.manual_seed(0)
img, gt = torch.rand(3, H, W, device=dev), torch.rand(3, H, W, device=dev)
omap = torch.randn(3, H, W, device=dev)
view = torch.eye(4, device=dev)
theta, conf = torch.rand(H, W, device=dev) * 3.14, torch.rand(H, W, device=dev)
mask = torch.rand(H, W, device=dev) > 0.5
bg3 = torch.zeros(3, device=dev)


import ctypes as C
L = rt.lib()
dmaps = torch.empty((3, 3, H, W), device=dev)
pa = torch.empty((L.hgs_ssim_l1_num_blocks(3, H, W), 2), device=dev)
pb = torch.empty((L.hgs_orientation_loss_num_blocks(H, W), 2), device=dev)
mask_u8 = mask.to(torch.uint8)
bg_c = (C.c_float * 3)(0.0, 0.0, 0.0)
win = fused.gaussian_window11()


def a():
    rt.check(L.hgs_ssim_l1_forward(rt.current_stream(), 3, H, W, win, rt.ptr(img), rt.ptr(gt), rt.ptr(dmaps), rt.ptr(pa)))


def b():
    rt.check(L.hgs_orientation_loss_forward(rt.current_stream(), H, W, rt.ptr(omap), rt.ptr(view), bg_c, 1e-7, rt.ptr(theta), rt.ptr(conf),
                                            rt.ptr(mask_u8), rt.ptr(pb)))


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both_serial():
    a(); b()


def both_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a()
    with torch.cuda.stream(s2):
        b()
    cur.wait_stream(s1); cur.wait_stream(s2)


print(f"eager: ssim {timed(a):.1f} us, per-pixel {timed(b):.1f} us, both on one stream {timed(both_serial):.1f} us, on two streams {timed(both_streams):.1f} us")
for name, fn in (("one stream", both_serial), ("fork / join", both_streams)):
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cs):
            for _ in range(20):
                fn()
    print(f"graph of 20 x ({name}): {timed(g.replay, 50) / 20:.1f} us per pair")
