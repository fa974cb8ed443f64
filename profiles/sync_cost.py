import time, torch
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
def cost():
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts)//2] * 1e6
print("idle synchronize, no extra streams: %.1f us" % cost())
streams = []
for n in (1, 2, 4, 8, 16):
    while len(streams) < n:
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            x.add_(1.0)
        streams.append(s)
    print("idle synchronize, %d extra streams: %.1f us" % (n, cost()))
