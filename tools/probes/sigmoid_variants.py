"""Probe: which float expression reproduces torch.sigmoid bit for bit (build: hipcc --offload-arch=gfx950 -shared -fPIC
-O3 sigmoid_variants.hip -o /tmp/sig.so, here; run on the GPU box)."""
import ctypes as C, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_sig.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-O3", os.path.join(here, "sigmoid_variants.hip"), "-o", so])
L = C.CDLL(so)
L.sig_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
n = 1 << 20
x = torch.randn(n, device="cuda") * 6
out = torch.empty(8, n, device="cuda")
L.sig_run(C.c_void_p(torch.cuda.current_stream().cuda_stream), n, x.data_ptr(), out.data_ptr())
torch.cuda.synchronize()
ref = torch.sigmoid(x)
names = ["1/(1+expf)", "rcp(1+expf)", "1/(1+__expf)", "rcp(1+__expf)", "fdividef(expf)", "fdividef(__expf)", "1/(1+exp2f(x*log2e))", "rcp(1+v_exp(x*log2e))"]
for k, nm in enumerate(names):
    d = (out[k].view(torch.int32) - ref.view(torch.int32)).abs()
    print(f"{nm:28s} mismatching {int((d != 0).sum()):8d}  max ulp {int(d.max())}")
