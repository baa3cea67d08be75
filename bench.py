#!/usr/bin/env python3
"""bench.py — zstd block-decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload full_4a|huf_literals|raw_rle|full_4b|mix]

A "step" is one pass of the hot path (cz_decode_batch_device: one persistent-grid launch of
cz_decode_frames_kernel) over one batch of synthetic frames that already sits in HBM.
Default workload = BASELINE config 4a: 10 000 single-block frames per GPU, each a full
compressed 128 KiB block (Huffman 4-stream literals + 32 768 FSE-coded sequences + match copy).
Frames shard by rank with no data-path collective (weak scaling: 10 000 frames per GPU).

Prints ONE JSON line on rank 0 (see the driver contract in the task description), with
  roofline      achieved = algorithmic bytes / mean kernel duration (hipEvents around the kernel
                on the stream it runs on), against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference algorithm) timed on a bounded sample of
                the same frames on this host's cores (rank 0, N=1 only)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="full_4a", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU")
    ap.add_argument("--gather", action="store_true", help="include an RCCL gather of the decoded arenas to rank 0 in the step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the extra per-config measurements (N=1)")
    ap.add_argument("--verify", type=int, default=16, help="frames per rank checked against the oracle after the run")
    ap.add_argument("--no-chain-prepass", action="store_true", help="run the FSE chains inside cz_decode_frames_kernel (single launch)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real runs); gloo only to rehearse the N>1 code path on a box with fewer GPUs than ranks")
    args = ap.parse_args()

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
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")     # device of the tiny control tensors
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import cairo_zstd_amd as cz
    from cairo_zstd_amd import dist as czdist
    from cairo_zstd_amd import synth

    # ---- synthetic batch for this rank (frames rank*F .. rank*F+F-1 of the global batch)
    F = args.frames
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.time()
    batch = synth.generate(args.workload, F, first_index=czdist.shard_first_index(F, rank), nthreads=max(1, min(32, ncpu // max(1, min(world, 8)))))
    gen_s = time.time() - t0
    out_off, out_cap, out_total = batch.out_layout(256)
    alg_bytes = int(batch.length.sum() + batch.regen.sum())       # compressed bytes read once + decoded bytes written once
    regen_bytes = int(batch.regen.sum())

    t_in = torch.from_numpy(batch.base).to(dev)
    t_off = torch.from_numpy(batch.off.astype(np.int64)).to(dev)
    t_len = torch.from_numpy(batch.length.astype(np.int64)).to(dev)
    t_ooff = torch.from_numpy(out_off.astype(np.int64)).to(dev)
    t_ocap = torch.from_numpy(out_cap.astype(np.int64)).to(dev)
    t_out = torch.empty(out_total, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(F * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    ctx = cz.Context(local_dev, stream.cuda_stream)
    # the FSE-chain pre-pass pays for long chains in large blocks (configs 4a/4b); on short, irregular
    # blocks (mix) and on blocks without sequences it is measured slower than in-kernel chains
    chain_prepass = not args.no_chain_prepass and args.workload in ("full_4a", "full_4b")
    if chain_prepass:
        ctx.set_chain_arena(int(batch.length.sum()) * 6 + (64 << 20))      # 8 B per sequence + 32 B per block

    gather_bufs = None
    if args.gather and world > 1:
        sizes = czdist.all_sizes(out_total, cdev)
        if rank == 0:
            gather_bufs = [torch.empty(sz, dtype=torch.uint8, device=cdev) for sz in sizes]

    def step():
        ctx.decode_batch_device(t_in.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), F, t_out.data_ptr(),
                                t_ooff.data_ptr(), t_ocap.data_ptr(), t_res.data_ptr())
        if args.gather and world > 1:
            czdist.gather_to_root(t_out if args.dist_backend == "nccl" else t_out.cpu(), gather_bufs, 0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # Per-launch kernel durations for the roofline: K more launches of the same step, each read
    # from the hipEvent pair the library records around the kernel on the stream it runs on
    # (reading a pair needs a sync, which must stay out of the timed region above).
    kernel_ms, chain_ms = [], []
    for _ in range(args.steps):
        ctx.decode_batch_device(t_in.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), F, t_out.data_ptr(),
                                t_ooff.data_ptr(), t_ocap.data_ptr(), t_res.data_ptr())
        kernel_ms.append(ctx.last_kernel_ms())
        chain_ms.append(ctx.last_chain_ms())
    torch.cuda.synchronize()

    # ---- correctness gate: every frame OK + sizes; a sample bit-exact against the oracle
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    ok = bool((res["status"] == 0).all() and (res["bytes_produced"] == batch.regen).all())
    verified = 0
    if args.verify:
        import oracle
        out_host = t_out.cpu().numpy()
        for i in np.linspace(0, F - 1, num=min(args.verify, F), dtype=np.int64):
            st, ref, _ = oracle.decode_frame(batch.frame(int(i)), cap=int(batch.regen[i]) + 16)
            got = out_host[int(out_off[i]): int(out_off[i] + batch.regen[i])].tobytes()
            ok = ok and st == 0 and got == ref
            verified += 1
        del out_host

    if world > 1:
        elapsed = czdist.max_over_ranks(elapsed, cdev)
        regen_all, alg_all, ok_cnt = czdist.sum_over_ranks([regen_bytes, alg_bytes, 1.0 if ok else 0.0], cdev)
        ok_all = int(ok_cnt) == world
    else:
        regen_all, alg_all, ok_all = float(regen_bytes), float(alg_bytes), ok

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
        for wl in ("raw_rle", "huf_literals", "full_4a"):
            if wl == args.workload:
                continue
            # the chain pre-pass only pays for frames that have sequences sections
            ctx.set_chain_arena(int(batch.length.sum()) * 6 + (64 << 20) if wl in ("full_4a", "full_4b") and not args.no_chain_prepass else 0)
            ob = synth.generate(wl, F, nthreads=max(1, min(32, ncpu)))
            o_off, o_cap, o_total = ob.out_layout(256)
            ti = torch.from_numpy(ob.base).to(dev)
            td = [torch.from_numpy(x.astype(np.int64)).to(dev) for x in (ob.off, ob.length, o_off, o_cap)]
            to = torch.empty(o_total, dtype=torch.uint8, device=dev)
            tr = torch.zeros(F * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            ms = []
            for it in range(3):
                ctx.decode_batch_device(ti.data_ptr(), td[0].data_ptr(), td[1].data_ptr(), F, to.data_ptr(), td[2].data_ptr(),
                                        td[3].data_ptr(), tr.data_ptr())
                ms.append(ctx.last_kernel_ms())
            r2 = tr.cpu().numpy().view(cz.RESULT_DTYPE)
            okw = bool((r2["status"] == 0).all() and (r2["bytes_produced"] == ob.regen).all())
            ab = int(ob.length.sum() + ob.regen.sum())
            k = float(np.mean(ms[1:]))
            others[wl] = {"decompressed_MBps": float(ob.regen.sum()) / (k * 1e-3) / 1e6, "kernel_ms": k,
                          "algorithmic_GBps": ab / (k * 1e-3) / 1e9, "roofline_frac": ab / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "all_frames_ok": okw}
            del ti, td, to, tr

    # ---- the same batch with a content checksum on every frame, XXH64 computed and
    # compared inside cz_decode_frames_kernel (N=1 only, after the timed region)
    cksum_leg = None
    if world == 1 and not args.no_other_workloads:
        try:
            import xxhash
        except ImportError:
            xxhash = None
        if xxhash is not None:
            out_host = t_out.cpu().numpy()
            n_len = batch.length.astype(np.int64) + 4
            n_off = np.concatenate(([0], np.cumsum((n_len + 15) & ~15)[:-1])).astype(np.int64)
            nb = np.zeros(int(n_off[-1] + n_len[-1]) + 16, dtype=np.uint8)
            for i in range(F):
                o, l, d = int(batch.off[i]), int(batch.length[i]), int(n_off[i])
                nb[d:d + l] = batch.base[o:o + l]
                nb[d + 4] |= 4                                            # Content_Checksum_flag
                h = xxhash.xxh64_intdigest(out_host[int(out_off[i]): int(out_off[i] + batch.regen[i])]) & 0xFFFFFFFF
                nb[d + l:d + l + 4] = np.frombuffer(h.to_bytes(4, "little"), np.uint8)
            del out_host
            ti = torch.from_numpy(nb).to(dev)
            tno, tnl = torch.from_numpy(n_off).to(dev), torch.from_numpy(n_len).to(dev)
            ctx.set_chain_arena(int(batch.length.sum()) * 6 + (64 << 20) if chain_prepass else 0)
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

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "decompressed MB/s (whole node), 128 KiB-block batch",
            "value": regen_all * args.steps / elapsed / 1e6,
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]}", "frames_per_gpu": F,
                       "frames_total": F * world, "compressed_bytes_per_gpu": int(batch.length.sum()),
                       "decoded_bytes_per_gpu": regen_bytes, "parallelism": f"frames sharded over {world} GPU(s), no data-path collective",
                       "gather_in_step": bool(args.gather and world > 1),
                       "launches_per_step": "cz_chain_kernel + cz_decode_frames_kernel" if chain_prepass else "cz_decode_frames_kernel"},
            "bit_exact": bool(ok_all), "frames_verified_vs_oracle": verified,
            "algorithmic_GBps_whole_job": alg_all * args.steps / elapsed / 1e9,
            "roofline": {"bound": "hbm", "kernel": "cz_chain_kernel + cz_decode_frames_kernel (one step)" if chain_prepass else "cz_decode_frames_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_mean": k_ms,
                         "kernel_ms_all": [round(float(x), 4) for x in kernel_ms],
                         "chain_kernel_ms_mean": float(np.mean(chain_ms)), "decode_kernel_ms_mean": k_ms - float(np.mean(chain_ms)),
                         **ctx.launch_info()},
            "synth_seconds": round(gen_s, 2),
        }
        # HBM traffic from the PMC counters is collected in separate rocprofv3 --pmc passes (FETCH_SIZE,
        # WRITE_SIZE) of this same command and committed under profiles/: bench.py cannot read PMCs itself
        pmc = os.path.join(ROOT, "profiles", "r1", "pmc_hbm_traffic_full_4a.json")
        if args.workload == "full_4a" and chain_prepass and F == 10000 and os.path.exists(pmc):
            t = json.load(open(pmc))
            line["roofline"]["traffic"] = t["fetch_bytes_uncorrected"] + t["write_bytes"]
            line["roofline"]["traffic_source"] = "profiles/r1/pmc_hbm_traffic_full_4a.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH_SIZE uncorrected: narrow reads), bytes per step, both kernels"
        if copy_ceiling is not None:
            line["roofline"]["empirical_copy_GBps"] = copy_ceiling      # torch device-to-device copy on this box, read+write
            line["roofline"]["frac_of_empirical_copy"] = achieved / copy_ceiling
        if others:
            line["other_workloads"] = others
        if cksum_leg:
            line["with_content_checksum_verified_on_device"] = cksum_leg
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            threads = ncpu
            # bounded sample: grow the frame count until the budget is used
            n_s = min(F, max(threads * 2, 64))
            sample_s, done, passes, cpu_regen = 0.0, 0, 0, 0.0
            t_begin = time.perf_counter()
            while True:
                idx = np.arange(done, min(F, done + n_s))
                t1 = time.perf_counter()
                _, olen, ost = oracle.decode_batch(batch.base, batch.off[idx], batch.length[idx], out_off[idx] - out_off[idx[0]],
                                                   out_cap[idx], int(out_off[idx[-1]] + out_cap[idx[-1]] - out_off[idx[0]]) + 256,
                                                   nthreads=threads)
                sample_s += time.perf_counter() - t1
                assert (ost == 0).all()
                cpu_regen += float(batch.regen[idx].sum())
                done += idx.size
                if done >= F:                                           # many host cores: go over the batch again until the budget is used
                    done, passes = 0, passes + 1
                if time.perf_counter() - t_begin > args.cpu_seconds:
                    break
                n_s = min(F - done, n_s * 2) if passes == 0 else F
            frames_done = passes * F + done
            line["cpu_baseline"] = {"value": cpu_regen / sample_s / 1e6, "unit": "MB/s", "cores": threads, "kind": "port",
                                    "sample": f"{frames_done} frame decodes ({passes} full passes over the same {F}-frame batch + {done} frames), oracle/zstd_oracle.c "
                                              f"(C restatement of the reference), {threads} pthreads, one frame per task, {sample_s:.2f} s wall = {sample_s * threads:.0f} core-seconds"}
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
