import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gorilla-tsc_amd")]
import numpy as np, torch
import pygtsc as G
from pygtsc import synth
P = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
for S in [int(x) for x in sys.argv[1].split(",")]:
    b = synth.random_walk(S=S, P=P) if len(sys.argv) > 3 else synth.readme_workload(S=S, P=P)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    d_block, d_ts, d_val = d(b.block_ts), d(b.ts), d(b.val)
    slot = G.max_words(P)
    d_out = torch.zeros(S * slot, dtype=torch.int64, device="cuda"); d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    ctx = G.Context(0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(8):
        t0 = time.perf_counter()
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, uniform_npoints=P, uniform_slot_words=slot); ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print("S=%d P=%d encode %.3f ms" % (S, P, best * 1e3))
    ctx.close()
    del d_out, d_ts, d_val
    torch.cuda.empty_cache()
