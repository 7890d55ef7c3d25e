"""Experiment: ONE headline batch handed to K scanners as K newline-aligned sub-batches in flight (submit_device / wait, one stream
each, no fork) against the forked single scan. python tools/split_exp.py [K ...]"""
import sys, time
sys.path.insert(0, ".")
import torch
import matchy_amd as M
from tools import synth
cfg = synth.config("c2")
db = M.Database(synth.build_db(cfg))
lines = 10_000_000
cap = lines * 200 + (1 << 20)
host = torch.empty(cap, dtype=torch.uint8)
n = synth.make_log_into(cfg, 0, lines, host.data_ptr(), cap)
hv = host[:n].numpy()
dev = torch.device("cuda", 0)
whole = torch.empty(n + 64, dtype=torch.uint8, device=dev)
whole[:n].copy_(host[:n])
torch.cuda.synchronize()
sc0 = M.Scanner(db, device=0)
def full():
    r = sc0.scan_device(whole.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream, fetch_mode=1)
    out = (r.lines, r.candidates, r.n_hits); r.close(); return out
for _ in range(10): ref = full()
t = time.perf_counter()
for _ in range(30): full()
print(f"forked single scan: {(time.perf_counter() - t) / 30 * 1e3:.4f} ms", ref, flush=True)
for K in [int(a) for a in sys.argv[1:]] or [2, 3, 4]:
    for shares in ([1] * K, list(range(K, 0, -1))):
        tot = sum(shares)
        cuts = [0]
        acc = 0
        for s in shares[:-1]:
            acc += s
            p = int(n * acc / tot)
            while hv[p - 1] != 10: p += 1
            cuts.append(p)
        cuts.append(n)
        bufs = []
        for a, b in zip(cuts, cuts[1:]):
            d = torch.empty(b - a + 64, dtype=torch.uint8, device=dev)
            d[:b - a].copy_(host[a:b])
            bufs.append((d, b - a))
        torch.cuda.synchronize()
        scs = [M.Scanner(db, device=0) for _ in range(K)]
        sts = [torch.cuda.Stream(device=dev) for _ in range(K)]
        def split():
            for i in range(K):
                scs[i].submit_device(bufs[i][0].data_ptr(), bufs[i][1], stream=sts[i].cuda_stream, fetch_mode=1)
            tot = [0, 0, 0]
            for i in range(K):
                r = scs[i].wait()
                tot[0] += r.lines; tot[1] += r.candidates; tot[2] += r.n_hits
                r.close()
            return tuple(tot)
        for _ in range(10): got = split()
        t = time.perf_counter()
        for _ in range(30): split()
        print(f"K={K} shares={shares}: {(time.perf_counter() - t) / 30 * 1e3:.4f} ms per whole batch, same counts: {got == ref}", flush=True)
        for s in scs: s.close()
