#!/usr/bin/env python3
"""Headline benchmark: BLS12-381 pairings/sec (Miller loop + final exponentiation), batch 2^20 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One process per GPU.  With --gpus N > 1 and no torch.distributed environment, this process starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`, BEFORE
any GPU call) and relays rank 0's JSON line and the children's exit status; started under torchrun it is a rank.

A step = one pass of the hot path (bls381_multi_pairing_batch_dev, k = 1) over this rank's batch of 2^20 synthetic
(G1, G2) pairs already resident in HBM; pairings are independent, so ranks share nothing on the data path (weak scaling,
total batch = N * 2^20 -- BASELINE.json config "2^23 across 8 GPUs").  Rank 0 prints ONE JSON line.

Inputs: pools of 4096 subgroup points built on the GPU with the library's own
point kernels; element i pairs P_{i mod 4096} with Q_{(i * 2654435761 >> 12) mod 4096} -- 2^20 distinct pairings per rank.

`roofline` prices the kernel against HBM as BASELINE.json asks (the path is integer-ALU bound, so that fraction is tiny by
construction; `int_roofline` prices it against the measured v_mad_u64_u32 issue rate).  `cpu_baseline` times the C
restatement of the same path (oracle/, a port -- the Rust reference cannot be built here) on this box's host cores, on a
bounded sample.  oracle/ is touched only for that leg and for the parity spot check, both outside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "python"))

# algorithmic figures (DESIGN.md "Measurement")
BYTES_PER_PAIRING = 96 + 192 + 576          # affine G1 + affine G2 in, Fq12 out, each touched once
# oracle op counts (`python oracle/pyref.py`: Fp multiplications, squarings counted as multiplications) -- the ALGORITHMIC unit
# of work of SURVEY 8d, not what the kernels execute (they run fewer multiplications: homogeneous steps, compressed squarings,
# the shorter hard part; and more multiply-adds per multiplication: 392 in the 28-bit representation)
FP_MUL_MILLER, FP_MUL_FEXP, FP_MUL_MILLER_K3 = 6916, 8491, 16284
FP_MUL_MODE_A = 11903                       # Mode A with the strength reductions the kernel applies; 20 655 with the reference's literal generic products
FP_MUL_PER_PAIRING = FP_MUL_MILLER + FP_MUL_FEXP
FP_MUL_PER_UNIT = {"pairing": FP_MUL_MILLER + FP_MUL_FEXP, "miller": FP_MUL_MILLER, "ref_opt": FP_MUL_MODE_A,
                   "groth16": FP_MUL_MILLER_K3 + FP_MUL_FEXP, "groth16_prepared": FP_MUL_MILLER_K3 + FP_MUL_FEXP}
MAC_PER_FP_MUL = 288                        # 32x32->64 limb MACs per 381-bit Montgomery multiplication (2 * 12^2)
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s
INT_MAC_PEAK = 30.3e12                      # measured v_mad_u64_u32 lane-ops/s, profiles/r01_valu_rates_mi355x.txt
SPOT_CHECK = 256                            # elements of rank 0's batch compared with the oracle (spread over the batch)
SPOT_SEED = 20261005                        # seed of the sample of checked elements
EXCHANGE_DEADLINE_S = 300                   # N > 1: the scatter / launch / gather flow after the timed region gets this long


def baseline_metric():
    """BASELINE.json's metric string, verbatim (falls back to the same text if the file is absent)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "BLS12-381 pairings/sec (Miller+final exp), batch=2^20, 1/2/4/8 MI355X"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--workload", choices=["pairing", "miller", "ref_opt", "groth16", "groth16_prepared"], default="pairing")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--pool", type=int, default=4096, help="points per synthetic pool (SURVEY 8d: 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--from-host", action="store_true", help="also time the host-pointer entry point (bls381_multi_pairing_batch: "
                    "H2D + kernels + D2H as a caller with host buffers sees it) and report it in config.from_host; never `value`")
    ap.add_argument("--hom-miller", type=int, choices=[-1, 0, 1], default=-1, help="BLS381_OPT_PAIRING_HOMOGENEOUS_MILLER (-1 = library default)")
    ap.add_argument("--fexp-pipelines", type=int, default=0, help="BLS381_OPT_FEXP_PIPELINES (1..4; 0 = library default)")
    ap.add_argument("--fexp-mode", type=int, choices=[-1, 0, 1, 2, 3], default=-1,
                    help="BLS381_OPT_FEXP_MODE: 0 self-contained kernel, 1 batched inversion, 2 staged (compressed squarings), "
                         "3 auto (staged from 2^10 elements); -1 = library default (auto)")
    ap.add_argument("--gather", action="store_true", help="after the timed region, collect every rank's results on rank 0 with "
                    "bls381_amd.sharding.gather_outputs (RCCL all_gather over xGMI) and report its time; never part of `value`")
    ap.add_argument("--exchange", type=int, choices=[0, 1], default=-1, help="after the timed region, run the one-rank-owns-the-batch "
                    "flow once: scatter of the inputs from rank 0, local launch, gather of the results (config.exchange); untimed by "
                    "`value`.  Default: on for N > 1")
    ap.add_argument("--single-process", action="store_true", help="after the timed region, time the one-process / several-devices "
                    "form a Rust caller would use (bls381_create_multi + bls381_multi_pairing_batch_sharded over every visible device, "
                    "host buffers in, host buffers out) and report it in config.single_process; never `value`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """--gpus N > 1 outside torchrun: start the N ranks as children (no GPU call has happened in this process) and relay."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print("bench.py: the %d-rank run failed (exit %d, %d result lines)" % (args.gpus, proc.returncode, len(lines)), file=sys.stderr)
        return proc.returncode or 1
    print(lines[0], flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    distributed = world > 1
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_index = local_rank if args.backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    import bls381_amd as B
    from bls381_amd import synthetic

    n = 1 << args.batch_log2
    eng = B.Engine(dev_index)
    # an explicit stream: every launch of a step, the HIP events that time it and the copies of the exchange flow are
    # ordered on it (torch's default stream is the legacy null stream)
    stream = torch.cuda.Stream(device=dev)
    eng.set_stream(stream.cuda_stream)
    if args.block:
        eng.set_option(B.OPT_BLOCK_SIZE, args.block)
    if args.fexp_mode >= 0:
        eng.set_option(B.OPT_FEXP_MODE, args.fexp_mode)
    if args.fexp_pipelines:
        eng.set_option(B.OPT_FEXP_PIPELINES, args.fexp_pipelines)
    if args.hom_miller >= 0:
        eng.set_option(B.OPT_PAIRING_HOMOGENEOUS_MILLER, args.hom_miller)
    g16 = args.workload in ("groth16", "groth16_prepared")
    k = 3 if g16 else 1
    n_units = n // 4 if g16 else n          # 2^18 proofs x 3 pairs (BASELINE config 4)

    jac = args.workload == "ref_opt"
    pools = synthetic.build_pools(eng, args.pool)                # on the GPU, with the library's point kernels
    first = rank * n_units                                       # ranks own consecutive element ranges of one global batch
    if jac:
        g1, g2 = synthetic.jacobian_batch(eng, pools, n_units, first)
    else:
        g1, g2 = synthetic.affine_batch(pools, n_units, k, first)
    dprep = None
    if args.workload == "groth16_prepared":
        # Q1, Q2 fixed for the batch (the verifying key's points): keep Q0 per proof, prepare the other two once
        g2all = g2.reshape(n_units, 3, 24)
        g2_fixed = np.ascontiguousarray(g2all[0, 1:3]).reshape(-1)
        prep, _ = eng.g2_prepare(g2_fixed)
        g2 = np.ascontiguousarray(g2all[:, 0]).reshape(-1)
        dprep = torch.from_numpy(prep.view(np.int64)).to(dev)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev)
    d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    dout = torch.empty(n_units * 72, dtype=torch.int64, device=dev)
    dfl = torch.empty(n_units, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        if args.workload == "groth16_prepared":
            eng.groth16_pairing_dev(d1.data_ptr(), d2.data_ptr(), dprep.data_ptr(), dout.data_ptr(), dfl.data_ptr(), n_units)
        elif args.workload in ("pairing", "groth16"):
            eng.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), k, dout.data_ptr(), dfl.data_ptr(), n_units)
        elif args.workload == "miller":
            eng.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), k, dout.data_ptr(), dfl.data_ptr(), n_units)
        else:
            eng.optimized_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), dout.data_ptr(), dfl.data_ptr(), n_units)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms, miller_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(eng.last_kernel_ms())      # HIP events on the launch stream, recorded by the library
        miller_ms.append(eng.last_phase_ms(0))      # ... and one between the Miller kernel and the final exponentiation
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_ms = None
    if args.gather and distributed and args.backend == "nccl":
        from bls381_amd.sharding import gather_outputs
        torch.cuda.synchronize(); dist.barrier()
        g0 = time.perf_counter()
        full = gather_outputs(dout, n_units * world, 72, dst=0)
        torch.cuda.synchronize(); dist.barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert full.numel() == n_units * world * 72
        del full

    exchange, exchange_hung = None, False
    do_exchange = args.exchange == 1 or (args.exchange == -1 and distributed)
    if do_exchange and distributed and args.workload in ("pairing", "groth16"):
        # SURVEY 8e flow for a caller whose batch lives on rank 0: scatter inputs, launch, gather results (bls381_amd.sharding).
        # Rank 0 rebuilds the whole global batch (ranks own consecutive element ranges of it), every rank then checks the
        # scattered-and-recomputed results against what it computed from its resident shard.
        # The flow runs on a worker thread with a deadline: the timed region is over and its numbers are final -- a point-to-point
        # exchange that never completes (it has not yet run on a real multi-GPU node) must cost this record, not the headline line.
        import threading
        box = {}

        def exchange_flow():
            from bls381_amd.sharding import scatter_batch, gather_results
            host = args.backend != "nccl"                     # gloo rehearsal: point-to-point needs CPU tensors
            xdev = torch.device("cpu") if host else dev
            n_total = n_units * world
            full_in = None
            if rank == 0:
                f1, f2 = synthetic.affine_batch(pools, n_total, k, 0)
                full_in = [torch.from_numpy(f1.view(np.int64)).to(xdev), torch.from_numpy(f2.view(np.int64)).to(xdev)]
            torch.cuda.synchronize(); dist.barrier()
            x0 = time.perf_counter()
            a, b = scatter_batch(full_in, [12 * k, 24 * k], n_total, src=0, device=xdev)
            torch.cuda.synchronize(); dist.barrier()
            x1 = time.perf_counter()
            a, b = a.to(dev), b.to(dev)
            xo = torch.empty(n_units * 72, dtype=torch.int64, device=dev); xf = torch.empty(n_units, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            eng.multi_pairing_dev(a.data_ptr(), b.data_ptr(), k, xo.data_ptr(), xf.data_ptr(), n_units)
            eng.sync()
            torch.cuda.synchronize(); dist.barrier()
            x2 = time.perf_counter()
            same_local = bool(torch.equal(xo, dout))
            res = gather_results([xo.to(xdev), xf.to(xdev)], [72, 1], n_total, dst=0)
            torch.cuda.synchronize(); dist.barrier()
            x3 = time.perf_counter()
            ok_t = torch.tensor([1 if same_local else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            if rank == 0:
                same = bool(ok_t.item()) and bool(torch.equal(res[0][: n_units * 72].to(dev), dout))
                if not same:
                    raise SystemExit("bench: scattered/gathered results differ from the resident-shard results")
                in_b, out_b = n_total * 288 * k * (world - 1) / world, n_total * 577 * (world - 1) / world
                return {"scatter_ms": (x1 - x0) * 1e3, "compute_ms": (x2 - x1) * 1e3, "gather_ms": (x3 - x2) * 1e3,
                        "scatter_GBps": in_b / (x1 - x0) / 1e9, "gather_GBps": out_b / (x3 - x2) / 1e9,
                            "end_to_end_units_per_s": n_total / (x3 - x0), "staged_through_host": host,
                            "results_match_resident_shards": same}
            return None

        def worker():
            try:
                torch.cuda.set_device(dev_index)
                box["result"] = exchange_flow()
            except SystemExit as e:
                box["fatal"] = str(e)
            except (RuntimeError, ValueError) as e:      # an exchange that fails the same way on every rank must not cost the headline line
                box["result"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        th = threading.Thread(target=worker, daemon=True)
        th.start()
        th.join(EXCHANGE_DEADLINE_S)
        if th.is_alive():
            exchange_hung = True
            exchange = {"error": "no completion within %d s (ranks exit without tearing the process group down)" % EXCHANGE_DEADLINE_S}
        elif "fatal" in box:
            raise SystemExit(box["fatal"])
        else:
            exchange = box.get("result")

    # spot parity inside the bench, EVERY workload: SPOT_CHECK elements spread over rank 0's batch against the oracle (checker only,
    # after the timed region); config.parity_spot_check_cover says how many different pool points P and Q the checked elements use.
    ok, spot_cover = None, None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from coracle import COracle
        # a seeded sample without replacement (a fixed stride aliases with the pool-index hash: stride n/256 met only P_0, stride
        # 4099 only 72 different Q), always with the first and the last element
        idx = np.sort(np.random.default_rng(SPOT_SEED).choice(n_units, size=min(SPOT_CHECK, n_units), replace=False).astype(np.int64))
        idx[0], idx[-1] = 0, n_units - 1
        w1, w2 = (18, 36) if jac else (12 * k, 24 * k)
        s1 = np.ascontiguousarray(g1.reshape(n_units, w1)[idx]).reshape(-1)
        if args.workload == "groth16_prepared":       # the oracle walks all three curves: variable Q0 of the proof + the two fixed points
            s2 = np.concatenate([g2.reshape(n_units, 24)[idx], np.broadcast_to(g2_fixed.reshape(1, 48), (len(idx), 48))], axis=1).reshape(-1)
        else:
            s2 = np.ascontiguousarray(g2.reshape(n_units, w2)[idx]).reshape(-1)
        orc = COracle()
        if jac:
            exp, _ = orc.miller_ref_opt(s1, s2, threads=8)
        else:
            exp, _ = orc.multi(s1, np.ascontiguousarray(s2), k, final_exp=args.workload != "miller", threads=8)
        got = dout.view(n_units, 72)[torch.from_numpy(idx).to(dev)].cpu().numpy().view(np.uint64).reshape(-1)
        ok = bool(np.array_equal(got, exp))
        if not ok:
            raise SystemExit("bench: HIP results differ from the oracle -- number would be invalid")
        pi, qi = synthetic.element_indices(n_units * k, args.pool, first * k)
        sel = (idx[:, None] * k + np.arange(k)[None, :]).reshape(-1)
        selq = idx * k if args.workload == "groth16_prepared" else sel      # prepared: only pair 0 has a per-proof Q
        spot_cover = {"distinct_P": int(len(np.unique(pi[sel]))), "distinct_Q": int(len(np.unique(qi[selq])))}

    from_host = None
    if args.from_host and rank == 0 and args.workload == "pairing":
        hout = np.zeros(n_units * 72, dtype=np.uint64)                  # a caller's result buffer, touched once and re-used
        eng.multi_pairing(g1, g2, 1, out=hout)                          # warm: staging buffers, page faults of the buffers
        hs = []
        for _ in range(max(2, args.steps)):
            h0 = time.perf_counter()
            eng.multi_pairing(g1, g2, 1, out=hout)
            hs.append(time.perf_counter() - h0)
        if not np.array_equal(hout.view(np.int64), dout.cpu().numpy()):
            raise SystemExit("bench: host-pointer results differ from the device-pointer results")
        best = min(hs)
        from_host = {"ms_per_step": best * 1e3, "pairings_per_s": n_units / best, "vs_resident": (elapsed / args.steps) / best,
                     "note": "bls381_multi_pairing_batch with pageable numpy buffers the caller re-uses: uploads, kernels and downloads "
                             "included, pieces of 2^18 elements double-buffered on the context's copy streams"}

    # One host process, every visible device (bls381_create_multi + bls381_multi_pairing_batch_sharded): the shape the reference's
    # caller -- a single Rust program -- would use.  Host buffers in and out, contiguous ranges, one host thread per device.
    single_process = None
    if args.single_process and rank == 0 and world == 1 and args.workload in ("pairing", "groth16", "groth16_prepared"):
        me = B.MultiEngine(None)
        ndv = me.device_count
        hout = np.zeros(n_units * 72, dtype=np.uint64)

        def sp_run():
            if args.workload == "groth16_prepared":
                hout[:] = me.groth16_pairing(g1, g2, prep)[0]
            else:
                me.multi_pairing(g1, g2, k, out=hout)
        sp_run()                                                        # warm
        hs = []
        for _ in range(max(2, args.steps)):
            h0 = time.perf_counter()
            sp_run()
            hs.append(time.perf_counter() - h0)
        if not np.array_equal(hout.view(np.int64), dout.cpu().numpy()):
            raise SystemExit("bench: single-process sharded results differ from the device-pointer results")
        best = min(hs)
        single_process = {"devices": ndv, "ms_per_batch": best * 1e3, "units_per_s": n_units / best, "batch": n_units,
                          "note": "bls381_multi_pairing_batch_sharded / bls381_groth16_pairing_batch_sharded: ONE batch of %d units cut into %d contiguous ranges (strong scaling "
                                  "inside one process), pageable host buffers, PCIe copies included; results equal to the resident run" % (n_units, ndv)}
        me.close()

    # HBM traffic per launch from the last committed PMC passes (tools/collect_profile.sh + summarize_profile.py):
    # separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, read side x2 (gfx950).
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get("workloads", {}).get("%s@%d" % (args.workload, args.batch_log2))
            if ent is None and args.workload == "pairing" and n == (1 << 20) and "traffic_bytes_per_launch" in tj:
                ent = tj
            if ent is not None:
                traffic = float(ent["traffic_bytes_per_launch"])
                traffic_src = "%s @ %s" % (ent.get("tag", "?"), ent.get("git", "?"))
        except Exception:
            traffic = None

    if rank == 0:
        total_units = n_units * world * args.steps
        value = total_units / elapsed
        avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
        bytes_per_unit = BYTES_PER_PAIRING if k == 1 else (3 * 288 + 576)
        if jac:
            bytes_per_unit = 144 + 288 + 576
        achieved_gbs = bytes_per_unit * n_units / avg_kernel_s / 1e9
        # what the library ran for the last call, from the library (bls381_get_option, BLS381_INFO_*)
        hom = bool(eng.get_option(B.INFO_LAST_MILLER_HOMOGENEOUS))
        fexp_mode, fexp_pipes = eng.get_option(B.INFO_LAST_FEXP_MODE), eng.get_option(B.INFO_LAST_FEXP_PIPELINES)
        fexp_kernels = {B.FEXP_SELF_CONTAINED: "k_final_exp", B.FEXP_BATCHED_INVERSION: "k_fexp_norm + k_fp_batch_inv + k_final_exp_given",
                        B.FEXP_STAGED: "k_fexp_norm_park + 5 x (k_fp_batch_inv + k_fexp_glue*_wide + k_fexp_chain) + k_fp_batch_inv + "
                                       "k_fexp_stage6_wide, %d part-batch pipeline%s" % (fexp_pipes, "s" if fexp_pipes != 1 else "")}.get(fexp_mode, "none")
        miller_kernel = {"pairing": "k_multi_miller<1,%s>" % ("true" if hom else "false"), "groth16": "k_multi_miller<3,%s>" % ("true" if hom else "false"),
                         "groth16_prepared": "k_prepared_miller<1,2,%s>" % ("true" if hom else "false"), "miller": "k_multi_miller<1,false>",
                         "ref_opt": "k_miller_ref_opt_wide"}[args.workload]
        has_fexp = args.workload in ("pairing", "groth16", "groth16_prepared")
        line = {
            "metric": baseline_metric() if args.workload == "pairing" else "BLS12-381 %s units/sec" % args.workload,
            "value": value, "unit": "pairings/s" if args.workload == "pairing" else ("proofs/s" if g16 else "miller loops/s"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "batch 2^%d full pairings (Miller loop + Fp12 final exponentiation), 1xMI355X per rank"
                                   % args.batch_log2 if args.workload == "pairing" else args.workload,
                       "batch_per_gpu": n_units, "global_batch": n_units * world, "pairs_per_unit": k,
                       "inputs": "pools of %d subgroup points built on the GPU; element i = (P[i mod %d], Q[(i*2654435761>>12) mod %d])%s"
                                 % (args.pool, args.pool, args.pool, ", per-element random Jacobian representative" if jac else ""),
                       "io": "arkworks Montgomery 6xu64 AoS, resident in HBM",
                       "parity_spot_check": ok, "parity_spot_check_elements": len(idx) if ok is not None else 0,
                       "parity_spot_check_cover": spot_cover,
                       "gather_ms": gather_ms, "exchange": exchange, "from_host": from_host, "single_process": single_process},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE; committed passes of this command: "
                                         "profiles/latest_traffic.json, %s)" % traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_unit * n_units,
                         "kernel": miller_kernel + (" + %s (one step = all of these launches)" % fexp_kernels if has_fexp else ""),
                         "kernel_ms_avg": avg_kernel_s * 1e3, "bytes_per_unit": bytes_per_unit,
                         "note": "integer-ALU bound path: HBM fraction is tiny by construction, see int_roofline"},
        }
        fpm = FP_MUL_PER_UNIT[args.workload]
        macs = fpm * MAC_PER_FP_MUL * n_units / avg_kernel_s
        line["int_roofline"] = {"bound": "valu_int32_mad", "achieved": macs / 1e12, "peak": INT_MAC_PEAK / 1e12,
                                "unit": "T limb-MAC/s", "frac": macs / INT_MAC_PEAK,
                                "fp_mul_per_unit": fpm, "mac_per_fp_mul": MAC_PER_FP_MUL,
                                "note": "algorithmic: the oracle's Fp-multiplication count x 288 limb-MACs (SURVEY 8d), not the executed instruction mix"}
        if has_fexp and min(miller_ms) > 0:      # the dominant kernel on its own: its launch duration (HIP events) against its own unit of work
            mk = float(np.mean(miller_ms)) * 1e-3
            mfp = FP_MUL_MILLER if k == 1 else FP_MUL_MILLER_K3
            line["roofline"]["dominant_kernel"] = {
                "kernel": miller_kernel, "ms_avg": mk * 1e3,
                "hbm_GBps": bytes_per_unit * n_units / mk / 1e9,
                "int_frac": mfp * MAC_PER_FP_MUL * n_units / mk / INT_MAC_PEAK, "int_frac_kind": "algorithmic (oracle op count)",
                "final_exponentiation_ms_avg": (avg_kernel_s - mk) * 1e3}
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from coracle import COracle
            orc = COracle()
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            per_core = {"pairing": 768, "miller": 1536, "ref_opt": 512, "groth16": 384, "groth16_prepared": 384}[args.workload]
            sample = per_core * threads       # ~15 s of CPU work (about 1.2 ms per pairing and core on the GPU box)
            w1, w2 = (18, 36) if jac else (12 * k, 24 * k)
            cg1 = g1[: sample * w1]
            cg2 = synthetic.affine_batch(pools, sample, k, first)[1] if args.workload == "groth16_prepared" else g2[: sample * w2]

            def cpu_run(a, b, th):
                if jac:
                    return orc.miller_ref_opt(a, b, threads=th)
                return orc.multi(a, b, k, final_exp=args.workload != "miller", threads=th)
            cpu_run(cg1[: w1 * threads], cg2[: w2 * threads], threads)      # warm
            c0 = time.perf_counter()
            cpu_run(cg1, cg2, threads)
            cdt = time.perf_counter() - c0
            one = 256                                                       # the single-thread figure of SURVEY 8d
            c1 = time.perf_counter()
            cpu_run(cg1[: one * w1], cg2[: one * w2], 1)
            cdt1 = time.perf_counter() - c1
            line["cpu_baseline"] = {"value": sample / cdt, "unit": line["unit"], "cores": threads, "kind": "port",
                                    "single_thread": one / cdt1,
                                    "sample": "the first %d units of the same synthetic batch, C restatement oracle/bls381_oracle.c "
                                              "(same algorithm; the Rust reference cannot be built here), %.1f s wall on %d threads; "
                                              "single_thread: the first %d units on one thread, %.2f s" % (sample, cdt, threads, one, cdt1)}
        print(json.dumps(line), flush=True)
    if distributed and exchange_hung:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)                          # a collective is stuck: destroy_process_group would wait for it
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
