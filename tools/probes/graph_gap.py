"""Probe: GPU-visible cost of launching a captured HIP graph back to back -- one executable graph replayed N times
against two executable graphs of the same work replayed alternately (does the runtime wait for the previous launch of the
SAME executable graph?), and against the same kernels launched eagerly."""
import sys, time, torch
n_k = int(sys.argv[1]) if len(sys.argv) > 1 else 14      # kernels per graph
work = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22  # elements per kernel (~20 us each at 4M floats)
x = torch.zeros(work, device="cuda")
def body():
    for _ in range(n_k):
        x.add_(1.0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body(); torch.cuda.synchronize()
    graphs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
        graphs.append(g)
    def timed(fn, n=300):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n): fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6
    e = timed(lambda i=0: body())
    a = timed(lambda i=0: graphs[0].replay())
    b = timed(lambda i=0: graphs[i & 1].replay())
print(f"{n_k} kernels of {work} floats: eager {e:.1f} us/iter, one exec {a:.1f} us/iter, two execs alternating {b:.1f} us/iter")
