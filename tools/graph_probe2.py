import sys, torch
mode = sys.argv[1]; kind = sys.argv[2]
w = torch.randn(1000, 16, device="cuda", requires_grad=True)
e = torch.zeros(1000, 0, 3, device="cuda", requires_grad=True)
opt = torch.optim.Adam([w] + ([e] if "empty" in kind else []), lr=torch.tensor(0.01, device="cuda") if "adam" in kind else 0.01, capturable=True, fused=True)
x = torch.randn(64, 1000, device="cuda")
def body():
    l = (x @ w).pow(2).mean() + (torch.cat((w[:, :1, None].expand(-1, 1, 3), e), dim=1).sum() if "empty" in kind else 0)
    l.backward()
    if "adam" in kind: opt.step()
    return l
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        body(); opt.zero_grad(set_to_none=True)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print("capturing", mode, kind, flush=True)
with torch.cuda.graph(g, capture_error_mode=mode):
    out = body()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed OK", float(out), flush=True)
