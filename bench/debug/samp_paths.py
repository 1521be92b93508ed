import sys, torch
sys.path.insert(0, "tests")
from test_sampler_full_gpu import draw
from distributed_llm_inference_amd import ops
K = ops.load()
g = torch.Generator().manual_seed(3)
z = torch.randn(128256, generator=g) * 2
a, pa, ida = draw(K, z, 1.0, 1024, 0.95, 4096, dbg=8)
b, pb, idb = draw(K, z, 1.0, 1025, 0.95, 4096, dbg=8)
order = torch.argsort(z, descending=True)
rank = torch.empty_like(order); rank[order] = torch.arange(order.numel())
ra, rb = rank[a.long()], rank[b.long()]
d = (a != b).nonzero().flatten()
print("differ", d.numel(), "rank diffs (a-b) histogram:", torch.unique((ra - rb)[d], return_counts=True))
print("a ranks", ra[d][:16].tolist()); print("b ranks", rb[d][:16].tolist())
print("sorted path dbg", pa[0].tolist(), ida[0].tolist()); print("full path dbg", pb[0].tolist(), idb[0].tolist())
p = torch.softmax(z[order[:1025]].double(), 0); print("cum at 1024", float(p[:1024].sum()), "top probs", p[:4].tolist())

import numpy as np
def philox_u(seed, step):
    c = [step, 0, 0, 0]; k = [seed & 0xffffffff, seed >> 32]
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]; p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xffffffff, p1 & 0xffffffff, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xffffffff, p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return (c[0] >> 8) / 16777216.0
for kk, got in ((1024, a), (1025, b)):
    zs = z[order[:kk]].double(); e = torch.exp(zs - zs[0]); cum = e.cumsum(0); tot = cum[-1]
    prev = cum - e; keep = (prev / tot <= 0.95); nkeep = int(keep.sum()); kt = cum[nkeep - 1]
    exp_rank = torch.tensor([int((cum[:nkeep] <= philox_u(s, 0) * kt).sum()) for s in range(4096)])
    gr = rank[got.long()]
    print("k", kk, "nkeep", nkeep, "match expected:", float((gr == exp_rank).float().mean()), "diff hist", torch.unique(gr - exp_rank, return_counts=True))
