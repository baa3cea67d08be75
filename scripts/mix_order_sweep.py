import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import cairo_zstd_amd as cz
from cairo_zstd_amd import synth
n = 12500
b = synth.generate("mix", n, nthreads=16)
dev = torch.device("cuda:0")
ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
def run(label, idx, lit):
    off, ln, rg = b.off[idx], b.length[idx], b.regen[idx]
    cap = rg.astype(np.uint64); pad = (cap + np.uint64(255)) // np.uint64(256) * np.uint64(256)
    ooff = np.zeros(idx.size, dtype=np.uint64); ooff[1:] = np.cumsum(pad[:-1])
    t = [torch.from_numpy(x).to(dev) for x in (b.base, off.astype(np.int64), ln.astype(np.int64), ooff.astype(np.int64), cap.astype(np.int64))]
    t_out = torch.empty(int(pad.sum()), dtype=torch.uint8, device=dev)
    t_res = torch.zeros(idx.size * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ctx.set_chain_arena(int(ln.sum()) * 8 + (64 << 20), min_sequences=0)
    ctx.set_literal_arena(int(rg.sum()) + (16 << 20) if lit else 0)
    tot, ch = [], []
    for it in range(4):
        ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), idx.size, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
        tot.append(ctx.last_kernel_ms()); ch.append(ctx.last_chain_ms())
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    print(f"{label:34s} total {np.mean(tot[1:]):7.3f} chain {np.mean(ch[1:]):6.3f} lit tail {ctx.last_literals_tail_ms():6.3f} counts {ctx.last_prepass_counts(idx.size)} ok={bool((res['status']==0).all())}", flush=True)
order = np.argsort(-b.regen.astype(np.int64))
byc = np.argsort(-b.length.astype(np.int64))
for lit in (0, 1):
    run(f"index order lit={lit}", np.arange(n), lit)
    run(f"largest decoded first lit={lit}", order, lit)
    run(f"largest compressed first lit={lit}", byc, lit)
ctx.close()
