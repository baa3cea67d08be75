#!/usr/bin/env python3
"""bench.py — zstd block-decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload full_4a|huf_literals|raw_rle|full_4b|mix]

A "step" is one pass of the hot path (one cz_decode_batch_device call) over one batch of synthetic
frames that already sits in HBM.  The call launches
    cz_scan_kernel x2  ->  cz_chain_kernel || cz_huf1_kernel || cz_tile_kernel  (three streams: FSE chains, Huffman literals
                                                                                  and Raw / RLE runs, unit of work = a block)
                       ->  cz_huf_kernel (the literals cz_huf1_kernel did not get to beside the chain kernel)
                       ->  cz_execute_frames_kernel || cz_wexec_kernel (the frames the pre-pass prepared: a wave per frame with
                           match sources from HBM, side by side with a workgroup per frame with the block in an LDS window — on
                           batches whose far offsets outweigh the near ones, decided on the device)
                       ->  cz_decode_frames_kernel (the rest)
(--no-chain-prepass: the one persistent-grid cz_decode_frames_kernel of round 1).
Default workload = BASELINE config 4a: 10 000 single-block frames per GPU, each a full compressed
128 KiB block (Huffman 4-stream literals + 32 768 FSE-coded sequences + match copy).

--gpus N > 1 started by hand re-launches itself as N ranks under torch.distributed.run (a child
process, before anything touches the GPU) and fails loudly when the box has fewer GPUs; under the
driver's own torchrun launch the ranks find RANK/WORLD_SIZE in the environment.  Frames shard by
rank with no data-path collective (weak scaling: 10 000 frames per GPU); the mix is first dealt by
algorithmic bytes (one untimed all_to_all of compressed bytes), and --gather times the RCCL gather
of the decoded bytes to rank 0 as a separate leg.

Prints ONE JSON line on rank 0 (see the driver contract in the task description), with
  roofline      achieved = algorithmic bytes / mean duration of the step's kernels (hipEvents inside
                the library, on the streams the kernels run on), against the 8 TB/s HBM peak;
                chain_latency_floor = how long cz_chain_kernel's slots need for this batch at the
                measured minimum step latency (a property of its slot count, not of zstd);
                frac_dominant_kernel = the same bytes / the longest stage of the step (chain kernel or execute stage);
                issue_bound_ms = the VALU instructions of the step's kernels (SQ counters, profiles/r5/sq_<workload>.json) at the
                measured issue rate of a SIMD (profiles/r5/microbench_issue.txt) over all SIMDs; frac_of_issue_bound = that / the step;
                traffic = PMC bytes from profiles/r5 when that file was measured on
                these very kernel sources (kernel_source_hash), else null
  cpu_baseline  the CPU oracle (a port of the reference algorithm) on all host cores over the whole
                batch, on one thread over a bounded sample, and libzstd on one thread and on all
                cores (rank 0, N=1)
  bit_exact     the last launch decodes into a 0xA5-poisoned buffer and EVERY frame is compared
                with the oracle's output by XXH64 — for the headline workload and for every entry of
                other_workloads
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# What a SIMD of gfx950 issues, measured (profiles/r5/microbench_issue.txt, scripts/micro/issue.hip): with two or more waves one
# integer VALU instruction of the kinds these kernels are made of (bit-field ops, funnel shifts, DPP, selects, SGPR operands) per
# 4.2 clocks — the 2-clock rate holds for runs of plain VOP1 / VOP2 adds and logic ops on VGPRs only —, a lone wave one instruction
# of any kind per 6.0; SALU 4.2 per SIMD, beside the VALU stream; clock in those loops 2.36 GHz.
ISSUE = {"clk_per_valu": 4.2, "clk_per_valu_plain_runs": 2.25, "clk_per_instruction_lone_wave": 6.0, "clk_per_salu": 4.2, "clock_ghz": 2.36, "simds": 1024,
         "source": "profiles/r5/microbench_issue.txt"}

WORKLOADS = {
    "raw_rle": "config2: single-block frames, 50% Raw 131072 B random / 50% RLE 131072 B",
    "huf_literals": "config3: literals-only compressed blocks, 131072 literals, 4-stream Huffman, fresh table per block",
    "full_4a": "config4a: full compressed blocks, 32768 literals (Huffman 4-stream) + 32768 sequences (ll~1, ml=3), FSE-compressed LL/OF/ML tables, 131072 B out",
    "full_4b": "config4b: reference-shape blocks, 65536 sequences, ~224 KiB out (beyond the zstd block limit)",
    "mix": "config5: corpus-like mix of Raw/RLE/Compressed multi-block frames",
}


def _spawn_ranks(n: int, backend: str) -> int:
    """Re-launches this command as n ranks (one per GPU) under torch.distributed.run and returns
    the launcher's exit code.  Runs in a process that has not initialised the GPU (counting
    devices does not), as a CHILD process — never an exec."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} GPUs on this node, {have} visible (one rank per GPU over RCCL; "
                  f"use --dist-backend gloo only to rehearse the N>1 code path)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _kernel_source_hash() -> str:
    """sha256 over the kernel sources: profiles/ files carry it, so a PMC measurement is only quoted for the code it was made on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cairo_zstd_amd", "csrc")
    for name in ("czstd_kernels.hip", "czstd_chain.hip", "czstd_pre.hip", "czstd_wexec.hip", "czstd_host.hip", "czstd_types.h"):
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def _libzstd():
    """ZSTD_decompress from the host's libzstd.so.1 when there is one (a second, stronger CPU baseline on spec-valid configs)."""
    import ctypes
    for name in ("libzstd.so.1", "libzstd.so"):
        try:
            L = ctypes.CDLL(name)
            L.ZSTD_decompress.restype = ctypes.c_size_t
            L.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
            L.ZSTD_isError.restype = ctypes.c_uint
            L.ZSTD_isError.argtypes = [ctypes.c_size_t]
            L.ZSTD_versionNumber.restype = ctypes.c_uint
            return L
        except OSError:
            continue
    return None


def _real_frames(n: int, level: int = 3, distinct: int = 384):
    """n frames of 128 KiB as a real encoder leaves them: text-like, log-like and mixed binary / text data compressed by the box's
    libzstd at `level`.  `distinct` different inputs are made (Python builds them: the bound), the batch repeats them.
    Returns (frames, originals, number of distinct frames), or None without a libzstd (or when it fails to compress)."""
    import ctypes
    try:
        Z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return None
    Z.ZSTD_compress.restype = ctypes.c_size_t
    Z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    Z.ZSTD_isError.restype = ctypes.c_uint
    Z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    rng = np.random.default_rng(7)
    SZ = 131072
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(3000)]
    blob = bytes(rng.integers(0, 256, 4096, dtype=np.uint8))
    dst = ctypes.create_string_buffer(2 * SZ)
    frames, origs = [], []
    for i in range(min(n, distinct)):
        kind = i % 3
        if kind == 0:                                                   # running text: words by a skewed distribution
            idx = np.minimum(rng.integers(0, len(words), SZ // 4), rng.integers(0, len(words), SZ // 4))
            d = b" ".join(words[int(j)] for j in idx)[:SZ]
        elif kind == 1:                                                 # log-like records
            rec = bytearray()
            t = int(rng.integers(0, 10 ** 9))
            while len(rec) < SZ:
                t += int(rng.integers(1, 50))
                rec += b"%010d host%02d GET /api/v1/item/%06d status=%d bytes=%d\n" % (t, int(rng.integers(0, 40)), int(rng.integers(0, 50000)), (200, 200, 200, 404, 500)[int(rng.integers(0, 5))], int(rng.integers(100, 90000)))
            d = bytes(rec[:SZ])
        else:                                                           # records of text fields and binary fields
            rec = bytearray()
            while len(rec) < SZ:
                o = int(rng.integers(0, 4000))
                rec += b"{\"id\": %d, \"name\": \"%s\", \"blob\": \"" % (int(rng.integers(0, 10 ** 6)), words[int(rng.integers(0, len(words)))]) + blob[o:o + int(rng.integers(8, 64))].hex().encode() + b"\"}\n"
            d = bytes(rec[:SZ])
        d = d.ljust(SZ, b".")
        m = Z.ZSTD_compress(dst, 2 * SZ, d, len(d), level)
        if Z.ZSTD_isError(m):                                           # (an error code is a huge size_t: never slice with it)
            return None
        frames.append(dst.raw[:m])
        origs.append(d)
    k = len(frames)
    return [frames[i % k] for i in range(n)], [origs[i % k] for i in range(n)], k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="full_4a", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (default 10000; 12500 for mix = BASELINE config 5 at 8 GPUs)")
    ap.add_argument("--gather", action="store_true", help="also time the step with an RCCL gather of the decoded arenas to rank 0 (reported separately)")
    ap.add_argument("--no-balance", action="store_true", help="mix on N > 1: keep every rank's contiguous slice instead of dealing frames by algorithmic bytes")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the extra per-config measurements (N=1)")
    ap.add_argument("--no-verify-all", action="store_true", help="check only a sample of the frames against the oracle (default: every frame, by XXH64)")
    ap.add_argument("--no-chain-prepass", action="store_true", help="diagnostic: everything inside cz_decode_frames_kernel (single launch, as in round 1)")
    ap.add_argument("--no-literals-pass", action="store_true", help="diagnostic: no cz_huf_kernel / cz_huf1_kernel / cz_tile_kernel; literals and Raw / RLE blocks inside the decode kernel")
    ap.add_argument("--no-exec-kernel", dest="exec_kernel", action="store_false", help="diagnostic: pre-passed frames on cz_decode_frames_kernel too, not on cz_execute_frames_kernel")
    ap.add_argument("--no-wexec-kernel", dest="wexec_kernel", action="store_false", help="diagnostic: cz_execute_frames_kernel alone, never side by side with cz_wexec_kernel")
    ap.add_argument("--real-frames", type=int, default=16000, help="frames of the real_libzstd_l3 entry of other_workloads (made by the box's libzstd at run time; 0: skip)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real runs); gloo only to rehearse the N>1 code path on a box with fewer GPUs than ranks")
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 12500 if args.workload == "mix" else 10000

    # ---- --gpus N without a launcher: start the N ranks ourselves (torch.distributed.run, one
    # process per GPU) BEFORE anything in this process touches the GPU, and exit with their code.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(_spawn_ranks(args.gpus, args.dist_backend))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    ngpu = torch.cuda.device_count()
    if args.dist_backend == "nccl":
        assert local_rank < ngpu, f"rank {rank}: LOCAL_RANK {local_rank} but only {ngpu} GPU(s) visible"
    local_dev = local_rank % ngpu
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")     # device of the control tensors / exchanged bytes
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import cairo_zstd_amd as cz
    import oracle
    from cairo_zstd_amd import dist as czdist
    from cairo_zstd_amd import synth

    # ---- synthetic batch for this rank (frames rank*F .. rank*F+F-1 of the global batch)
    F = args.frames
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.time()
    batch = synth.generate(args.workload, F, first_index=czdist.shard_first_index(F, rank), nthreads=max(1, min(32, ncpu // max(1, min(world, 8)))))
    gen_s = time.time() - t0
    balanced = False
    if world > 1 and args.workload == "mix" and not args.no_balance:
        # BASELINE config 5: the global batch is dealt to the GPUs by algorithmic bytes (one exchange of compressed bytes, untimed)
        nb, noff, nlen, nregen, _ = czdist.rebalance_frames(batch.base, batch.off, batch.length, batch.regen, cdev)
        batch = synth.Batch(nb, noff, nlen, nregen)
        balanced = True
    F = batch.n
    out_off, out_cap, out_total = batch.out_layout(256)
    alg_bytes = int(batch.length.sum() + batch.regen.sum())       # compressed bytes read once + decoded bytes written once
    regen_bytes = int(batch.regen.sum())

    dev_base = czdist.rebalance_frames.last_device_base if balanced else None
    t_in = dev_base if dev_base is not None and dev_base.device == dev else torch.from_numpy(batch.base).to(dev)   # (dealt frames stay on the GPU the exchange put them on)
    t_off = torch.from_numpy(batch.off.astype(np.int64)).to(dev)
    t_len = torch.from_numpy(batch.length.astype(np.int64)).to(dev)
    t_ooff = torch.from_numpy(out_off.astype(np.int64)).to(dev)
    t_ocap = torch.from_numpy(out_cap.astype(np.int64)).to(dev)
    t_out = torch.empty(out_total, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(F * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    # the context gets a stream of torch's making (handle 0 would mean a stream of the context's own, not ordered with torch's work):
    # everything this script does between launches synchronises the device, and the gather leg below waits for this stream
    stream = torch.cuda.Stream(device=dev)
    ctx = cz.Context(local_dev, stream.cuda_stream)
    # the block-parallel pre-pass (cz_scan_kernel + cz_chain_kernel || cz_huf1_kernel || cz_tile_kernel + cz_huf_kernel) for every workload
    chain_prepass = not args.no_chain_prepass
    # the arenas, sized from the batch's own headers (cz_scan_kernel's counting pass; 8 B per sequence + 1 312 B per block with sequences,
    # literal bytes of the blocks with sequences) + slack — not from a rule of thumb
    torch.cuda.synchronize()
    arena_bytes, lit_bytes = ctx.measure_batch(t_in.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), F, t_ocap.data_ptr())
    arena_bytes += 8 << 20
    lit_bytes += 8 << 20
    if chain_prepass:
        ctx.set_chain_arena(arena_bytes)
        ctx.set_exec_kernel(args.exec_kernel)
        ctx.set_wexec_kernel(args.wexec_kernel)
        ctx.set_literal_arena(0 if args.no_literals_pass else lit_bytes)
    torch.cuda.synchronize()

    def decode():
        ctx.decode_batch_device(t_in.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), F, t_out.data_ptr(),
                                t_ooff.data_ptr(), t_ocap.data_ptr(), t_res.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn):
        for _ in range(args.warmup):
            step_fn()
        barrier()
        t_begin = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        barrier()
        return time.perf_counter() - t_begin

    elapsed = timed(decode)
    # Per-launch kernel durations for the roofline: K more launches of the same step, each read
    # from the hipEvent pair the library records around the kernel on the stream it runs on
    # (reading a pair needs a sync, which must stay out of the timed region above).
    kernel_ms, chain_ms, exec_ms, tail_ms, wexec_ms = [], [], [], [], []
    for _ in range(args.steps):
        decode()
        kernel_ms.append(ctx.last_kernel_ms())
        chain_ms.append(ctx.last_chain_ms())                              # scans + cz_chain_kernel
        exec_ms.append(ctx.last_exec_ms())                                # the execute stage: cz_execute_frames_kernel and, side by side with it, cz_wexec_kernel
        wexec_ms.append(ctx.last_wexec_ms())                              # how long cz_wexec_kernel ran of that
        tail_ms.append(ctx.last_literals_tail_ms())                       # how long cz_huf_kernel / cz_huf1_kernel / cz_tile_kernel ran on after cz_chain_kernel
    torch.cuda.synchronize()
    wexec_counts = ctx.last_wexec_counts()

    # ---- decode + gather (config 5's exchange step), timed separately
    gather_leg = None
    if args.gather and world > 1:
        sizes = czdist.all_sizes(out_total, cdev)
        gather_bufs = [torch.empty(sz, dtype=torch.uint8, device=cdev) for sz in sizes] if rank == 0 else None

        def decode_and_gather():
            decode()
            torch.cuda.current_stream().wait_stream(stream)              # the gather runs on torch's (and RCCL's) streams: behind this step's decode
            czdist.gather_to_root(t_out if args.dist_backend == "nccl" else t_out.cpu(), gather_bufs, 0)

        g_elapsed = czdist.max_over_ranks(timed(decode_and_gather), cdev)
        gather_leg = {"ms_per_step": g_elapsed / args.steps * 1e3, "gathered_bytes_per_step": int(sum(sizes) - sizes[0]),
                      "call": "torch.distributed.batch_isend_irecv: every rank isend()s its decoded arena straight to rank 0 (RCCL ncclSend/ncclRecv group over xGMI)"}
        del gather_bufs

    # ---- correctness gate: ONE more launch into a poisoned output buffer (a match that read its source before it was
    # written would now find 0xA5, not the previous pass's bytes); every frame OK + sizes; every frame bit-exact
    # against the CPU oracle by XXH64 (the oracle pass is also the cpu_baseline measurement)
    t_out.fill_(0xA5)
    t_res.zero_()
    torch.cuda.synchronize()
    decode()
    torch.cuda.synchronize()
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    ok = bool((res["status"] == 0).all() and (res["bytes_produced"] == batch.regen).all())
    out_host = t_out.cpu().numpy()
    verified, cpu_leg = 0, None
    threads = ncpu
    if not args.no_verify_all:
        t1 = time.perf_counter()
        ref, olen, ost = oracle.decode_batch(batch.base, batch.off, batch.length, out_off, out_cap, out_total, nthreads=threads)
        oracle_s = time.perf_counter() - t1
        ok = ok and bool((ost == 0).all() and (olen == batch.regen).all())
        for i in range(F):
            lo, hi = int(out_off[i]), int(out_off[i] + batch.regen[i])
            if oracle.xxh64(out_host[lo:hi]) != oracle.xxh64(ref[lo:hi]):
                ok = False
                break
            verified += 1
        cpu_leg = {"value": regen_bytes / oracle_s / 1e6, "unit": "MB/s", "cores": threads, "kind": "port",
                   "sample": f"one pass over this rank's whole batch ({F} frames, {regen_bytes / 1e6:.0f} MB decoded), oracle/zstd_oracle.c (C restatement of "
                             f"the reference), {threads} pthreads, one frame per task, {oracle_s:.2f} s wall"}
        del ref
    else:
        for i in np.linspace(0, F - 1, num=min(16, F), dtype=np.int64):
            st, r1, _ = oracle.decode_frame(batch.frame(int(i)), cap=int(batch.regen[i]) + 16)
            got = out_host[int(out_off[i]): int(out_off[i] + batch.regen[i])].tobytes()
            ok = ok and st == 0 and got == r1
            verified += 1

    if world > 1:
        elapsed = czdist.max_over_ranks(elapsed, cdev)
        regen_all, alg_all, ok_cnt, frames_all = czdist.sum_over_ranks([regen_bytes, alg_bytes, 1.0 if ok else 0.0, F], cdev)
        ok_all = int(ok_cnt) == world
        k_ms_all = czdist.max_over_ranks(float(np.mean(kernel_ms)), cdev)
    else:
        regen_all, alg_all, ok_all, frames_all, k_ms_all = float(regen_bytes), float(alg_bytes), ok, F, float(np.mean(kernel_ms))

    # ---- the other single-GPU BASELINE configs, measured after the timed region (N=1 only): same
    # step definition, fewer steps; reported under "other_workloads" so the line shows every config
    others = {}
    copy_ceiling = None
    if world == 1 and not args.no_other_workloads:
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        b2 = torch.empty_like(a)
        for _ in range(2):
            b2.copy_(a)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            b2.copy_(a)
        torch.cuda.synchronize()
        copy_ceiling = 2.0 * a.numel() * 5 / (time.perf_counter() - t1) / 1e9
        del a, b2
        for wl in ("raw_rle", "huf_literals", "full_4a", "full_4b", "mix"):
            if wl == args.workload:
                continue
            nf = 12500 if wl == "mix" else 10000
            # the chain pre-pass only pays for frames with long sequences sections
            ob = synth.generate(wl, nf, nthreads=max(1, min(32, ncpu)))
            pre = not args.no_chain_prepass
            o_off, o_cap, o_total = ob.out_layout(256)
            ti = torch.from_numpy(ob.base).to(dev)
            td = [torch.from_numpy(x.astype(np.int64)).to(dev) for x in (ob.off, ob.length, o_off, o_cap)]
            torch.cuda.synchronize()
            ab_, lb_ = ctx.measure_batch(ti.data_ptr(), td[0].data_ptr(), td[1].data_ptr(), nf, td[3].data_ptr())
            ctx.set_chain_arena(ab_ + (8 << 20) if pre else 0)
            ctx.set_literal_arena(lb_ + (8 << 20) if pre and not args.no_literals_pass else 0)
            to = torch.empty(o_total, dtype=torch.uint8, device=dev)
            tr = torch.zeros(nf * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            ms, cms_, ems_, wms_ = [], [], [], []
            for it in range(3):
                ctx.decode_batch_device(ti.data_ptr(), td[0].data_ptr(), td[1].data_ptr(), nf, to.data_ptr(), td[2].data_ptr(),
                                        td[3].data_ptr(), tr.data_ptr())
                ms.append(ctx.last_kernel_ms())
                if pre:
                    cms_.append(ctx.last_chain_ms()); ems_.append(ctx.last_exec_ms()); wms_.append(ctx.last_wexec_ms())
            wxc = ctx.last_wexec_counts() if pre else (0, 0, 0)
            # one more launch into a poisoned buffer, every frame compared with the oracle by XXH64 (as for the headline workload)
            to.fill_(0xA5)
            tr.zero_()
            torch.cuda.synchronize()
            ctx.decode_batch_device(ti.data_ptr(), td[0].data_ptr(), td[1].data_ptr(), nf, to.data_ptr(), td[2].data_ptr(),
                                    td[3].data_ptr(), tr.data_ptr())
            torch.cuda.synchronize()
            r2 = tr.cpu().numpy().view(cz.RESULT_DTYPE)
            okw = bool((r2["status"] == 0).all() and (r2["bytes_produced"] == ob.regen).all())
            exact, nver = None, 0
            if not args.no_verify_all:
                oh = to.cpu().numpy()
                oref, olen2, ost2 = oracle.decode_batch(ob.base, ob.off, ob.length, o_off, o_cap, o_total, nthreads=threads)
                exact = okw and bool((ost2 == 0).all() and (olen2 == ob.regen).all())
                for i in range(nf):
                    lo, hi = int(o_off[i]), int(o_off[i] + ob.regen[i])
                    if oracle.xxh64(oh[lo:hi]) != oracle.xxh64(oref[lo:hi]):
                        exact = False
                        break
                    nver += 1
                del oh, oref
            ab = int(ob.length.sum() + ob.regen.sum())
            k = float(np.mean(ms[1:]))
            others[wl] = {"frames": nf, "decompressed_MBps": float(ob.regen.sum()) / (k * 1e-3) / 1e6, "kernel_ms": k,
                          "algorithmic_GBps": ab / (k * 1e-3) / 1e9, "roofline_frac": ab / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "all_frames_ok": okw, "bit_exact": exact, "frames_verified_vs_oracle": nver}
            if pre and cms_:
                # how the device arranged the execute stage of this batch: frames cz_wexec_kernel finished (side by side on far-offset
                # batches; the few large frames of a near-offset batch), the rest on cz_execute_frames_kernel / its 8-waves build
                others[wl].update({"chain_kernel_ms": float(np.mean(cms_[1:])), "exec_stage_ms": float(np.mean(ems_[1:])), "wexec_kernel_ms": float(np.mean(wms_[1:])),
                                   "frames_finished_by_wexec_kernel": int(wxc[1]), "frames_handed_on_by_wexec_kernel": int(wxc[2])})
            del ti, td, to, tr
        # real encoder output (not a BASELINE config: what a user of the library decodes): frames made here by the box's libzstd,
        # every decoded frame compared with its original, the distinct frames also with the oracle
        real = _real_frames(args.real_frames) if args.real_frames > 0 and not args.no_chain_prepass else None
        if real is not None:
            rf, ro, distinct = real
            nf = len(rf)
            lens = np.array([len(f) for f in rf], dtype=np.int64)
            roff = np.zeros(nf, dtype=np.int64); roff[1:] = np.cumsum(lens[:-1])
            rbase = np.frombuffer(b"".join(rf) + b"\0" * 64, dtype=np.uint8)
            SZ = len(ro[0])
            o_off = np.arange(nf, dtype=np.int64) * SZ
            ti = torch.from_numpy(rbase.copy()).to(dev)
            td = [torch.from_numpy(x).to(dev) for x in (roff, lens, o_off, np.full(nf, SZ, dtype=np.int64))]
            torch.cuda.synchronize()
            ab_, lb_ = ctx.measure_batch(ti.data_ptr(), td[0].data_ptr(), td[1].data_ptr(), nf, td[3].data_ptr())
            ctx.set_chain_arena(ab_ + (8 << 20))
            ctx.set_literal_arena(lb_ + (8 << 20) if not args.no_literals_pass else 0)
            to = torch.empty(nf * SZ, dtype=torch.uint8, device=dev)
            tr = torch.zeros(nf * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ms, cms, ems = [], [], []
            for it in range(4):
                if it == 3:
                    to.fill_(0xA5); tr.zero_(); torch.cuda.synchronize()
                ctx.decode_batch_device(ti.data_ptr(), td[0].data_ptr(), td[1].data_ptr(), nf, to.data_ptr(), td[2].data_ptr(), td[3].data_ptr(), tr.data_ptr())
                ms.append(ctx.last_kernel_ms()); cms.append(ctx.last_chain_ms()); ems.append(ctx.last_exec_ms())
            torch.cuda.synchronize()
            r2 = tr.cpu().numpy().view(cz.RESULT_DTYPE)
            oh = to.cpu().numpy()
            okw = bool((r2["status"] == 0).all() and (r2["bytes_produced"] == SZ).all())
            exact = okw and all(oh[i * SZ:(i + 1) * SZ].tobytes() == ro[i] for i in range(nf))
            exact = exact and all(oracle.decode_frame(rf[i], cap=SZ + 16)[1] == ro[i] for i in range(distinct))
            ab = int(lens.sum()) + nf * SZ
            k = float(np.mean(ms[1:3]))
            others["real_libzstd_l3"] = {"frames": nf, "distinct_frames": distinct, "what": "128 KiB frames of text-like, log-like and record-like data compressed by the box's libzstd at level 3 (made at run time)",
                                         "compression_ratio": nf * SZ / float(lens.sum()), "decompressed_MBps": nf * SZ / (k * 1e-3) / 1e6, "kernel_ms": k,
                                         "chain_kernel_ms": float(np.mean(cms[1:3])), "exec_stage_ms": float(np.mean(ems[1:3])),
                                         "algorithmic_GBps": ab / (k * 1e-3) / 1e9, "roofline_frac": ab / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "all_frames_ok": okw, "bit_exact": bool(exact), "frames_verified_vs_originals": nf if exact else 0}
            del ti, td, to, tr, oh
        ctx.set_chain_arena(arena_bytes if chain_prepass else 0)
        ctx.set_literal_arena(lit_bytes if chain_prepass and not args.no_literals_pass else 0)

    # ---- the same batch with a content checksum on every frame, XXH64 computed and
    # compared inside cz_decode_frames_kernel (N=1 only, after the timed region)
    cksum_leg = None
    if world == 1 and not args.no_other_workloads:
        n_len = batch.length.astype(np.int64) + 4
        n_off = np.concatenate(([0], np.cumsum((n_len + 15) & ~15)[:-1])).astype(np.int64)
        nb = np.zeros(int(n_off[-1] + n_len[-1]) + 16, dtype=np.uint8)
        for i in range(F):
            o, l, d = int(batch.off[i]), int(batch.length[i]), int(n_off[i])
            nb[d:d + l] = batch.base[o:o + l]
            nb[d + 4] |= 4                                            # Content_Checksum_flag
            h = oracle.xxh64(out_host[int(out_off[i]): int(out_off[i] + batch.regen[i])]) & 0xFFFFFFFF
            nb[d + l:d + l + 4] = np.frombuffer(h.to_bytes(4, "little"), np.uint8)
        ti = torch.from_numpy(nb).to(dev)
        tno, tnl = torch.from_numpy(n_off).to(dev), torch.from_numpy(n_len).to(dev)
        ctx.set_verify_checksum(True)
        ms = []
        for it in range(4):
            ctx.decode_batch_device(ti.data_ptr(), tno.data_ptr(), tnl.data_ptr(), F, t_out.data_ptr(), t_ooff.data_ptr(),
                                    t_ocap.data_ptr(), t_res.data_ptr())
            ms.append(ctx.last_kernel_ms())
        ctx.set_verify_checksum(False)
        r2 = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
        want = cz.RESULT_FINISHED | cz.RESULT_HAS_CHECKSUM | cz.RESULT_CHECKSUM_COMPUTED | cz.RESULT_CHECKSUM_MATCH
        k = float(np.mean(ms[1:]))
        cksum_leg = {"kernel_ms": k, "decompressed_MBps": regen_bytes / (k * 1e-3) / 1e6,
                     "all_frames_ok_and_checksums_match": bool((r2["status"] == 0).all() and ((r2["flags"] & want) == want).all())}
        del ti, tno, tnl, nb
    del out_host

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        lit_pass = chain_prepass and not args.no_literals_pass
        launches = (("cz_scan_kernel x2 + (cz_chain_kernel || cz_huf1_kernel || cz_tile_kernel) + cz_huf_kernel + " if lit_pass else "cz_scan_kernel x2 + cz_chain_kernel + ")
                    + (("(cz_execute_frames_kernel || cz_wexec_kernel) + " if args.wexec_kernel and wexec_counts[1] else "cz_execute_frames_kernel + ") if args.exec_kernel and lit_pass else "") + "cz_decode_frames_kernel") if chain_prepass else "cz_decode_frames_kernel"
        line = {
            "metric": "decompressed MB/s (whole node), 128 KiB-block batch",
            "value": regen_all * args.steps / elapsed / 1e6,
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]}", "frames_per_gpu": args.frames,
                       "frames_total": int(frames_all), "compressed_bytes_rank0": int(batch.length.sum()),
                       "decoded_bytes_rank0": regen_bytes, "chain_arena_bytes": int(arena_bytes) if chain_prepass else 0, "literal_arena_bytes": int(lit_bytes) if chain_prepass and not args.no_literals_pass else 0,
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"
                                      + (", dealt by algorithmic bytes (one untimed all_to_all of compressed bytes)" if balanced else ""),
                       "launches_per_step": launches},
            "kernel_source_hash": _kernel_source_hash(), "chain_prepass": bool(chain_prepass), "exec_kernel": bool(args.exec_kernel), "wexec_kernel": bool(args.wexec_kernel),
            "literals_pass": bool(lit_pass),
            "bit_exact": bool(ok_all), "frames_verified_vs_oracle_rank0": verified,
            "verification": "last launch decoded into a 0xA5-poisoned buffer; every frame compared with the CPU oracle by XXH64" if not args.no_verify_all else "16-frame sample",
            "algorithmic_GBps_whole_job": alg_all * args.steps / elapsed / 1e9,
            "roofline": {"bound": "hbm", "kernel": launches + " (one step)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_mean": k_ms,
                         "kernel_ms_all": [round(float(x), 4) for x in kernel_ms],
                         "chain_kernel_ms_mean": float(np.mean(chain_ms)), "prepass_tail_ms_mean": float(np.mean(tail_ms)),
                         "exec_kernel_ms_mean": float(np.mean(exec_ms)),
                         "exec_stage": "cz_execute_frames_kernel (a wave per frame)" + (f" side by side with cz_wexec_kernel (a workgroup per frame; it ran {float(np.mean(wexec_ms)):.3f} ms of the stage and finished {wexec_counts[1]} of the {F} frames)" if wexec_counts[1] else ""),
                         "frac_dominant_kernel": alg_bytes / (max(float(np.mean(chain_ms)), float(np.mean(exec_ms)), 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "dominant_kernel": "cz_chain_kernel" if float(np.mean(chain_ms)) >= float(np.mean(exec_ms)) else "execute stage",
                         "decode_kernel_ms_mean": k_ms - float(np.mean(chain_ms)) - float(np.mean(tail_ms)) - float(np.mean(exec_ms)),
                         "slowest_rank_kernel_ms_mean": k_ms_all,
                         **ctx.launch_info()},
            "synth_seconds": round(gen_s, 2),
        }
        if args.workload in ("full_4a", "full_4b"):
            # a property of cz_chain_kernel's layout, not of zstd: 10 240 slots (40 per CU, LDS-bound) each run ONE block's chain, so a
            # batch of F blocks lasts ceil(F / 10 240) x (sequences per block) x (step latency); 32.3 ns is the dependent table
            # chase alone on one wave (profiles/r2/microbench_chain_step.txt, variant 6) — a finer split of a chain, or more
            # slots, would lower it
            nseq = 32768 if args.workload == "full_4a" else 65536
            step_ns, slots = 32.3, 10240
            rounds = -(-F // slots)
            line["roofline"]["chain_latency_floor"] = {"sequences_per_block": nseq, "chain_slots": slots, "rounds": rounds, "min_step_ns": step_ns,
                                                       "floor_ms": rounds * nseq * step_ns * 1e-6,
                                                       "source": "profiles/r2/microbench_chain_step.txt (variant 6: the dependent table chase alone)"}
        # The instruction-issue bound of the step: every VALU instruction its kernels execute (SQ_INSTS_VALU of a counter pass, stamped with
        # the kernel sources) at the rate a SIMD issues them, spread over all 1 024 SIMDs — what the step would take if nothing ever
        # waited for memory, for another wave, or for a kernel boundary.  The entropy configs are bound by this, not by HBM bytes.
        sqf = os.path.join(ROOT, "profiles", "r5", f"sq_{args.workload}.json")
        if os.path.exists(sqf) and world == 1:
            q = json.load(open(sqf))
            if q.get("kernel_source_hash") == _kernel_source_hash() and q.get("frames") == F:
                valu, salu = q["step_totals"].get("SQ_INSTS_VALU", 0.0), q["step_totals"].get("SQ_INSTS_SALU", 0.0)
                rate = ISSUE["simds"] * ISSUE["clock_ghz"] * 1e9 / ISSUE["clk_per_valu"]
                ib = valu / rate * 1e3
                line["roofline"]["issue_bound_ms"] = ib
                line["roofline"]["frac_of_issue_bound"] = ib / k_ms
                line["roofline"]["issue_bound"] = {"valu_instructions_per_step": valu, "salu_instructions_per_step": salu, "valu_per_second_chip": rate,
                                                   "per_kernel_valu": {k: v.get("SQ_INSTS_VALU", 0.0) for k, v in q["per_kernel"].items() if v.get("SQ_INSTS_VALU", 0.0) >= 1e6},
                                                   **ISSUE, "counters": f"profiles/r5/sq_{args.workload}.json",
                                                   "note": "cz_chain_kernel runs one wave per SIMD (a chain step is a dependent instruction stream): its own floor is chain_latency_floor, and at 6.0 clocks per instruction its 31.5 instructions per step are the 93 ns it measures"}
            else:
                line["roofline"]["issue_bound_ms"] = None
                line["roofline"]["issue_bound_note"] = "profiles/r5 SQ counter file is from other kernel sources or another batch size: not quoted"
        # HBM traffic from the PMC counters is collected in separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
        # this same command and committed under profiles/ together with the hash of the kernel sources it was made on
        pmc = os.path.join(ROOT, "profiles", "r5", f"pmc_hbm_traffic_{args.workload}.json")
        if os.path.exists(pmc) and world == 1 and F == 10000:
            t = json.load(open(pmc))
            if t.get("kernel_source_hash") == _kernel_source_hash() and bool(t.get("chain_prepass")) == bool(chain_prepass) and bool(t.get("exec_kernel")) == bool(args.exec_kernel) and bool(t.get("wexec_kernel", True)) == bool(args.wexec_kernel):
                line["roofline"]["traffic"] = t["fetch_bytes_uncorrected"] + t["write_bytes"]
                alt = os.path.join(ROOT, "profiles", "r5", f"pmc_hbm_traffic_{args.workload}_alt.json")
                if os.path.exists(alt):
                    ta = json.load(open(alt))
                    if ta.get("kernel_source_hash") == _kernel_source_hash():
                        line["roofline"]["traffic_without_wexec_kernel"] = ta["fetch_bytes_uncorrected"] + ta["write_bytes"]
                        line["roofline"]["traffic_note"] = ("a counter pass runs a step's kernels one after the other: cz_wexec_kernel, first in line, then executes every frame it is listed "
                                                            "(traffic: records + literals in, output out); traffic_without_wexec_kernel is the same step with cz_execute_frames_kernel alone "
                                                            "(match sources from HBM).  Side by side, as timed, each kernel does its share of the frames: see roofline.exec_stage")
                line["roofline"]["traffic_source"] = f"profiles/r5/pmc_hbm_traffic_{args.workload}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on kernel sources {t['kernel_source_hash']}; FETCH_SIZE uncorrected), bytes per step, all kernels of the step"
            else:
                line["roofline"]["traffic_source"] = "profiles/r5 PMC file is from other kernel sources or launch options: not quoted"
        if copy_ceiling is not None:
            line["roofline"]["empirical_copy_GBps"] = copy_ceiling      # torch device-to-device copy on this box, read+write
            line["roofline"]["frac_of_empirical_copy"] = achieved / copy_ceiling
        if gather_leg:
            gather_leg["value_MBps_with_gather"] = regen_all / (gather_leg["ms_per_step"] * 1e-3) / 1e6
            line["with_gather_to_rank0"] = gather_leg
        if others:
            line["other_workloads"] = others
        if world == 1 and not args.no_other_workloads and chain_prepass:
            # two batches in flight: two contexts decode alternately without waiting for each other, in a process
            # of its own (it needs GPU_MAX_HW_QUEUES=8 before the runtime starts: six streams on four hardware queues run one after
            # the other).  A separate figure; `value` stays the one-batch-at-a-time rate.
            import subprocess
            try:
                pr = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pipelined.py"), args.workload, str(F), "2"], capture_output=True, timeout=300,
                                    env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
                js = [l[5:] for l in pr.stdout.decode().splitlines() if l.startswith("JSON ")]
                line["pipelined"] = json.loads(js[-1]) if js else {"error": pr.stderr.decode()[-300:]}
                if js:
                    line["pipelined"]["what"] = ("two contexts (arenas, streams and outputs of their own) decode the same batch alternately, enqueue only; ms per batch in the steady state. "
                                                 "On config 4a the second batch's kernels find no room beside cz_chain_kernel (135 KB of LDS per CU): nothing overlaps; on the corpus-like mix, "
                                                 "whose pre-pass leaves most of the chip idle, 8.4 -> 6.9 ms (profiles/r5/pipelined.txt)")
            except Exception as e:                                       # noqa: BLE001 (a diagnostic leg: never fail the bench line)
                line["pipelined"] = {"error": repr(e)[:300]}
        if cksum_leg:
            line["with_content_checksum_verified_on_device"] = cksum_leg
        if world == 1 and not args.no_cpu_baseline:
            if cpu_leg is None:
                t1 = time.perf_counter()
                _, olen, ost = oracle.decode_batch(batch.base, batch.off, batch.length, out_off, out_cap, out_total, nthreads=threads)
                s1 = time.perf_counter() - t1
                cpu_leg = {"value": regen_bytes / s1 / 1e6, "unit": "MB/s", "cores": threads, "kind": "port",
                           "sample": f"one pass over the whole batch ({F} frames), oracle/zstd_oracle.c, {threads} pthreads, {s1:.2f} s wall"}
            # single thread: a bounded prefix of the same batch
            n1, s1, done1 = 8, 0.0, 0
            budget = max(2.0, args.cpu_seconds * 0.25)
            while done1 < F and s1 < budget:
                idx = np.arange(done1, min(F, done1 + n1))
                t1 = time.perf_counter()
                _, olen, ost = oracle.decode_batch(batch.base, batch.off[idx], batch.length[idx], out_off[idx] - out_off[idx[0]], out_cap[idx],
                                                   int(out_off[idx[-1]] + out_cap[idx[-1]] - out_off[idx[0]]) + 256, nthreads=1)
                s1 += time.perf_counter() - t1
                done1 += idx.size
                n1 *= 2
            cpu_leg["single_thread"] = {"value": float(batch.regen[:done1].sum()) / s1 / 1e6, "unit": "MB/s", "cores": 1,
                                        "sample": f"the first {done1} frames of the batch, {s1:.2f} s"}
            Z = _libzstd()
            if Z is not None and args.workload in ("raw_rle", "huf_literals", "full_4a", "mix"):
                dst = np.empty(int(batch.regen.max()) + 64, dtype=np.uint8)
                zs, zdone, zbytes, zok = 0.0, 0, 0.0, True
                t_stop = time.perf_counter() + max(2.0, args.cpu_seconds * 0.25)
                while zdone < F and time.perf_counter() < t_stop:
                    o, l = int(batch.off[zdone]), int(batch.length[zdone])
                    t1 = time.perf_counter()
                    r = Z.ZSTD_decompress(dst.ctypes.data, dst.size, batch.base[o:o + l].ctypes.data, l)
                    zs += time.perf_counter() - t1
                    zok = zok and not Z.ZSTD_isError(r) and r == int(batch.regen[zdone])
                    zbytes += float(batch.regen[zdone])
                    zdone += 1
                cpu_leg["libzstd_single_thread"] = {"value": zbytes / zs / 1e6 if zs else None, "unit": "MB/s", "cores": 1, "all_frames_decoded": bool(zok),
                                                    "sample": f"ZSTD_decompress of libzstd {Z.ZSTD_versionNumber()} on the first {zdone} frames, {zs:.2f} s"}
                # all cores: the same libzstd on a pthread pool inside the oracle library (a Python thread per core spends its time on the GIL)
                zalls = []
                for _ in range(2):                                      # the first pass starts the threads and touches the output pages: the second one is quoted
                    t1 = time.perf_counter()
                    good = oracle.libzstd_batch(batch.base, batch.off, batch.length, out_off, out_cap, out_total, batch.regen, nthreads=threads)
                    zalls.append(time.perf_counter() - t1)
                zall = zalls[-1]
                if good >= 0:
                    cpu_leg["libzstd_all_cores"] = {"value": regen_bytes / zall / 1e6, "unit": "MB/s", "cores": threads, "all_frames_decoded": bool(good == F),
                                                    "sample": f"ZSTD_decompress of libzstd {Z.ZSTD_versionNumber()} over the whole batch ({F} frames) on {threads} pthreads, one frame per task: second of two passes, {zall:.2f} s wall (first: {zalls[0]:.2f} s)"}
            line["cpu_baseline"] = cpu_leg
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
