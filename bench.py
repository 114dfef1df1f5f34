#!/usr/bin/env python3
"""bench.py -- headline benchmark of the registration hot path on MI355X.

metric  : registration slice-pairs / s, whole job (BASELINE.json), one step = one training step of
          FlowNetS (forward + stn warps + OFEloss + backward + gradient all-reduce + Adam) on a
          batch of synthetic 256x256 pairs, 24 pairs per GPU (weak scaling), bf16 operands.
usage   : python bench.py --gpus N --steps K --warmup W      (N > 1 without RANK / WORLD_SIZE in the environment: bench.py
          starts its own N rank processes through torch.distributed.run before anything touches the GPU)
output  : ONE JSON line on rank 0 (see DESIGN.md section 7 for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
CONV_GFLOP_PER_PAIR_FWD = {"flownets": 10.114, "flownetc": 14.510, "pwc": 24.018}     # SURVEY section 8d (hook-measured on the reference)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_share():
    """CPUs this process may really use: cgroup quota if present, else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(B, size, steps, seed):
    """The CPU oracle (plain torch fp32 restatement of the reference path) timed on this host."""
    from oracle import nets, ops as oops
    from mireg.synth import make_pairs
    torch.manual_seed(seed)
    cores = min(cpu_share(), 64)
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads")
    model = nets.OpticalFlowReg("flownets")
    model.train()
    opt = torch.optim.Adam(model.parameters(), 1e-4, betas=(0.9, 0.999), eps=1e-4)
    x, _ = make_pairs(B, size, seed)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        flows, warped, _, _ = model(x)
        loss = oops.ofe_loss(flows, warped, x[:, 0:1])[3]
        opt.zero_grad()
        loss.backward()
        opt.step()
        log(f"cpu baseline step {i}: {time.perf_counter() - t0:.2f} s")
        if i > 0:
            times.append(time.perf_counter() - t0)
        if i >= 1 and sum(times) > 25:      # bounded sample
            break
    med = sorted(times)[len(times) // 2]
    return {"value": B / med, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} train steps (after 1 warm-up) of the CPU oracle, FlowNetS B={B} {size}x{size} fp32, median"}


def hbm_block(hbm):
    """Algorithmic HBM bytes / event-timed launch duration against 8 TB/s for the HBM-bound kernels of a step."""
    return {k: {"launches_per_step": v["launches"], "us_per_launch": round(v["ms"] * 1e3 / v["launches"], 1),
                "MB_per_launch": round(v["flops"] / v["launches"] / 1e6, 2),
                "GBps": round(v["flops"] / (v["ms"] * 1e-3) / 1e9, 1), "frac_of_8TBps": round(v["flops"] / (v["ms"] * 1e-3) / 8e12, 4)}
            for k, v in hbm.items()}


def model_leg(name, batch, precision, dev, steps=10, warm=4):
    """A short train-step measurement of another predictor (BASELINE configs[2] / [3]) on this GPU: throughput from a timed hipGraph
    region, then one eager step with per-launch events for the roofline blocks."""
    import mireg
    from mireg.engine import PROFILER
    from mireg.synth import make_pairs
    torch.manual_seed(6)
    model = mireg.opticalFlowReg(name, precision=precision).to(dev)
    tr = mireg.RegistrationTrainer(model, lr=1e-4, eps=1e-4, use_graph=True, autotune=True)
    x = make_pairs(batch, 256, seed=6)[0].to(dev)
    for _ in range(warm):
        tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tr.use_graph = False
    PROFILER.enabled, PROFILER.records, PROFILER.byte_records = True, [], []
    tr._fwd_bwd()
    tr._optim()
    summ, hbm = PROFILER.summary(), PROFILER.summary(bytes_=True)
    PROFILER.enabled = False
    fl = sum(v["flops"] for v in summ.values())
    ms = sum(v["ms"] for v in summ.values())
    dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
    peak = PEAK_TFLOPS[precision]
    out = {"pairs_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": batch, "loss_total": float(losses[3]),
           "step_tflops": round(3 * CONV_GFLOP_PER_PAIR_FWD[name] * batch / dt / 1e3, 1),
           "roofline": {"bound": "mfma", "kernel": dom[0], "achieved": round(dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12, 1), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12 / peak, 4), "traffic": None,
                        "all_contractions": {"achieved": round(fl / (ms * 1e-3) / 1e12, 1), "gflop_per_step": round(fl / 1e9, 1),
                                             "ms_per_step": round(ms, 3)}},
           "hbm_roofline": hbm_block(hbm)}
    del tr, model
    torch.cuda.empty_cache()
    return out


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes (one per GPU) as children BEFORE this
    process has made any GPU call, wait for them and leave with their exit code (never re-exec a process that touched the GPU)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 0.55 s timed region: the 20-step default of rounds 1-2 (56 ms) read 0.7 % faster than a sustained run
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=24, help="pairs per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--model", default="flownets")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-3d", action="store_true", help="skip the 128^3 affmodel leg")
    ap.add_argument("--no-other-models", action="store_true", help="skip the FlowNetC / PWC-DC-Net train-step legs")
    ap.add_argument("--tune-cache", default=None, help="JSON of measured launch shapes (written after tuning, reused when present)")
    ap.add_argument("--no-autotune", action="store_true", help="heuristic launch shapes (counter-collection runs: the tuning pass is slow there)")
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch pairs per GPU (the driver's contract); strong: --batch pairs in total, split over the ranks")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32 parity-mode train-step leg")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: initialise the process group, reduce one number, print the dist block "
                         "(runs without a GPU over gloo: tests/test_bench_contract.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if args.dry_run:
        backend = os.environ.get("MIREG_DIST_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo")
        if world > 1:
            if backend == "nccl":
                torch.cuda.set_device(local)
                torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                torch.distributed.init_process_group(backend)
            t = torch.tensor([float(rank)], device=(torch.device("cuda", local) if backend == "nccl" else "cpu"), dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            assert int(t.item()) == world - 1
            torch.distributed.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "scaling": args.scaling,
                              "dist": {"world": world, "backend": (torch.distributed.get_backend() if world > 1 else None)}}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.scaling == "strong":
        assert args.batch % world == 0, "strong scaling: --batch must divide by the number of ranks"
        args.batch //= world
    local = local % max(torch.cuda.device_count(), 1)            # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MIREG_DIST_BACKEND", "nccl")      # "gloo" only to rehearse the N>1 logic on one GPU
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    import mireg
    from mireg.engine import PROFILER
    from mireg.synth import make_pairs

    torch.manual_seed(6)
    model = mireg.opticalFlowReg(args.model, precision=args.precision).to(dev)
    if world > 1:  # identical replicas
        for p in model.parameters():
            torch.distributed.broadcast(p.data, 0)
        for b in model.buffers():
            torch.distributed.broadcast(b, 0)
    trainer = mireg.RegistrationTrainer(model, lr=1e-4, eps=1e-4, use_graph=not args.no_graph, autotune=not args.no_autotune,
                                        tune_cache=args.tune_cache)
    x_cpu, seg_cpu = make_pairs(args.batch, args.size, seed=6 + rank)
    x = x_cpu.to(dev)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model ready, warm-up")
    for _ in range(max(args.warmup, 3)):     # >= 3: two eager steps size the workspaces, then the graph is captured
        trainer.step(x)
    sync()
    log("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = trainer.step(x)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_vals = [float(v) for v in losses.tolist()]
    pairs = args.batch * world * args.steps

    # ---- roofline leg: per-launch timing of the MFMA contractions (events on the launch stream) ------------
    roof = None
    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step")
    if rank == 0:
        trainer.use_graph = False
        PROFILER.enabled, PROFILER.records, PROFILER.byte_records = True, [], []
        for _ in range(3):
            trainer._fwd_bwd()
            trainer._optim()
        summ = PROFILER.summary()
        hbm = PROFILER.summary(bytes_=True)
        PROFILER.enabled = False
        # same launches with nothing else in flight (no wgrad stream, no optimizer stream): the kernels' own rate
        trainer.eng.use_side_stream, trainer.overlap_optimizer = False, False
        PROFILER.enabled, PROFILER.records, PROFILER.byte_records = True, [], []
        for _ in range(2):
            trainer._fwd_bwd()
        iso = PROFILER.summary()
        PROFILER.enabled = False
        trainer.eng.use_side_stream, trainer.overlap_optimizer = True, True
        dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        tot_fl = sum(v["flops"] for v in summ.values())
        tot_ms = sum(v["ms"] for v in summ.values())
        peak = PEAK_TFLOPS[args.precision]
        ach = dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": dom[0], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": None,
                "avg_launch_us": round(dom[1]["ms"] * 1e3 / dom[1]["launches"], 2), "launches_per_step": dom[1]["launches"] // 3,
                "all_contractions": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), "ms_per_step": round(tot_ms / 3, 3),
                                     "gflop_per_step": round(tot_fl / 3 / 1e9, 1)},
                "isolated": {"note": "same launches, single stream (nothing overlapping)",
                             "kernel_tflops": round(iso[dom[0]]["flops"] / (iso[dom[0]]["ms"] * 1e-3) / 1e12, 1),
                             "kernel_frac": round(iso[dom[0]]["flops"] / (iso[dom[0]]["ms"] * 1e-3) / 1e12 / peak, 4),
                             "all_contractions_tflops": round(sum(v["flops"] for v in iso.values()) / (sum(v["ms"] for v in iso.values()) * 1e-3) / 1e12, 1),
                             "ms_per_step": round(sum(v["ms"] for v in iso.values()) / 2, 3),
                             "families": {k: {"ms_per_step": round(v["ms"] / 2, 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                                          for k, v in iso.items()}},
                "families": {k: {"launches_per_step": v["launches"] // 3, "ms_per_step": round(v["ms"] / 3, 3),
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in summ.items()},
                # HBM-bound kernels of the step: algorithmic bytes / event-timed duration vs 8 TB/s
                "hbm_kernels": {k: dict(v, launches_per_step=v["launches_per_step"] // 3) for k, v in hbm_block(hbm).items()}}

        # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process, so the figure comes from a
        # committed `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum` pass over this same command (scratch/pmc_bench.sh), per
        # launch, with the guide's gfx950 correction (wide reads are tallied at half their bytes: 2 * RDREQ * 64 B + WRREQ * 64 B).
        # The summary records the git revision of csrc/ it was taken on; when the kernels have changed since, or the dominant
        # kernel is not in it, traffic stays null rather than quoting a stale number.
        try:
            import subprocess
            prof = os.path.join(ROOT, "profiles", "round3_pmc_conv_kernels.json")
            pmc = json.load(open(prof))
            csrc = os.path.join(ROOT, "self-supervised-medical-image-registration-using-deep-optical-flow-estimation-with-brain-mri-data_amd", "csrc")
            import hashlib
            # the contraction kernels' sources (what the PMC pass measured): conv_*.hip + the shared header
            hsh = hashlib.sha1(b"".join(open(os.path.join(csrc, f), "rb").read() for f in sorted(os.listdir(csrc))
                                        if (f.startswith("conv_") and f.endswith(".hip")) or f == "mireg_common.h")).hexdigest()[:12]
            key = {"conv_wgrad_kernel<128,128>": "conv_wgrad_dma_kernel", "conv_wgrad_halo_kernel": "conv_wgrad_halo"}.get(dom[0], dom[0].split("<")[0])
            ents = [v for k, v in pmc["kernels"].items() if key in k]            # every instantiation of the kernel (e.g. its narrow-Cout layouts)
            ent = ents[0] if ents else None
            if ent is not None and pmc.get("csrc_sha1") == hsh and args.model == "flownets" and args.batch == 24 and args.size == 256:
                nl = sum(v["TCC_EA0_RDREQ_sum"]["launches"] for v in ents)
                rd = sum(v["TCC_EA0_RDREQ_sum"]["mean_per_launch"] * v["TCC_EA0_RDREQ_sum"]["launches"] for v in ents) / nl
                wr = sum(v["TCC_EA0_WRREQ_sum"]["mean_per_launch"] * v["TCC_EA0_WRREQ_sum"]["launches"] for v in ents) / nl
                roof["traffic"] = round(2 * rd * 64 + wr * 64)
                roof["traffic_note"] = f"bytes per launch, mean over the step's launches of {key}, from profiles/round3_pmc_conv_kernels.json (csrc sha1 {hsh})"
            else:
                roof["traffic_note"] = (f"null: profiles/round3_pmc_conv_kernels.json was taken on csrc sha1 {pmc.get('csrc_sha1')}, "
                                        f"this tree is {hsh}" if ent is not None else f"null: {key} not in the committed PMC summary")
        except Exception as e:                                       # noqa: BLE001 -- the bench line must still print
            roof["traffic_note"] = f"null: PMC summary unavailable ({e!r})"

    # ---- quality leg: warped Dice of the (random-init, K-step-trained) model, GPU vs CPU oracle ---------------
    dice = None
    if rank == 0:
        log("quality leg (Dice)")
        try:
            torch.set_num_threads(min(cpu_share(), 64))
            from oracle import nets as onets, ops as oops
            nb = 4
            xe, se = make_pairs(nb, args.size, seed=8, magnitude=(0.5, 1.0))
            ev = trainer.evaluate(xe.to(dev), se.to(dev))
            if "dice" not in ev:
                raise RuntimeError("top flow is not at image resolution: Dice undefined for this predictor")
            om = onets.OpticalFlowReg(args.model)
            om.load_state_dict(model.state_dict(), strict=False)
            om.eval()
            with torch.no_grad():
                _, _, wseg, _ = om(xe, se)
            d_cpu = [oops.dice_average(se[j, 0], wseg[j, 0]) for j in range(nb)]
            d_id = [oops.dice_average(se[j, 0], se[j, 1]) for j in range(nb)]
            if "dice" not in ev:
                raise RuntimeError("top flow is not at image resolution: Dice undefined for this predictor")
            dice = {"gpu_mean": round(float(ev["dice"].mean()), 5), "cpu_oracle_mean": round(sum(d_cpu) / nb, 5),
                    "unregistered_mean": round(sum(d_id) / nb, 5), "pairs": nb,
                    "note": "synthetic 4-label masks, weights after the timed steps (random init, no dataset)"}
        except Exception as e:  # the quality leg must never break the timing line
            dice = {"error": repr(e)}

    # ---- eval leg: forward + warps + OFEloss + warped-seg Dice per batch, eager, same weights ----------
    ev_line = None
    if rank == 0:
        try:
            xs, ss = x, seg_cpu.to(dev)
            for _ in range(2):
                trainer.evaluate(xs, ss)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_ev = 10
            for _ in range(n_ev):
                trainer.evaluate(xs, ss)
            torch.cuda.synchronize()
            t_ev = (time.perf_counter() - t0) / n_ev
            ev_line = {"pairs_per_s": round(args.batch / t_ev, 1), "ms_per_batch": round(t_ev * 1e3, 3),
                       "note": "inference.py:43-68 shaped: eval forward + stn + OFEloss + seg warp/round + per-sample Dice, 1 GPU, eager"}
            # the same batch with the per-sample metric block of inference.py:69-75 (MSE, PSNR, Pearson, mutual information)
            evm = trainer.evaluate(xs, ss, metrics=True)
            if "mse" in evm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_ev):
                    evm = trainer.evaluate(xs, ss, metrics=True)
                torch.cuda.synchronize()
                t_m = (time.perf_counter() - t0) / n_ev
                ev_line["with_metrics"] = {"pairs_per_s": round(args.batch / t_m, 1), "ms_per_batch": round(t_m * 1e3, 3),
                                           "mean_mse": float(evm["mse"].mean()), "mean_psnr": float(evm["psnr"].mean()),
                                           "mean_corr": float(evm["corr"].mean()), "mean_mi": float(evm["mi"].mean()),
                                           "mean_ssim": float(evm["ssim"].mean()) if "ssim" in evm else None}
            # on-device elastic deformation of the batch (the Rand2DElasticd step of the reference's CPU pipeline, dataset.py:78)
            from mireg.synth import elastic_deform
            gaug = torch.Generator().manual_seed(3)
            ctrl = ((torch.rand(args.batch, 2, args.size // 16 + 1, args.size // 16 + 1, generator=gaug) * 2 - 1) * 8).to(dev)
            fx, sg = xs[:, 0:1].contiguous(), ss[:, 0:1].contiguous().float()
            for _ in range(3):
                elastic_deform(fx, sg, ctrl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                elastic_deform(fx, sg, ctrl)
            torch.cuda.synchronize()
            t_a = (time.perf_counter() - t0) / 20
            ev_line["augment"] = {"pairs_per_s": round(args.batch / t_a, 1), "us_per_batch": round(t_a * 1e6, 1),
                                  "note": "bicubic control-grid field + bicubic image / nearest label resampling, eager"}
        except Exception as e:
            ev_line = dict(ev_line or {}, error=repr(e))

    # ---- 3-D leg (north_star "128^3 volumes"): the reference's only 3-D model, affmodel (models.py:156-191), forward +
    # affine grid / trilinear sampling + Affloss on synthetic 128^3 pairs, B = 8 (BASELINE configs[4] batch) ----------------
    vol_line = None
    if rank == 0 and not args.no_3d:
        def timed(fn, n, warm=2):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n, out

        try:
            g = torch.Generator(device="cpu").manual_seed(6)
            low = torch.rand(8, 2, 8, 8, 8, generator=g)
            vol = torch.nn.functional.interpolate(low, size=(128, 128, 128), mode="trilinear", align_corners=False).to(dev)
            aff = mireg.affmodel(fc_in=512 * 2 * 2 * 8, precision=args.precision).to(dev).eval()

            def aff_eval():
                with torch.no_grad():
                    para, wv = aff(vol)
                    return mireg.Affloss(wv, vol[:, 0:1])
            t3, l3 = timed(aff_eval, 5)
            vol_line = {"volumes_per_s": round(8 / t3, 1), "ms_per_batch": round(t3 * 1e3, 3), "batch": 8, "size": "128x128x128",
                        "loss": float(l3[2]), "note": "affmodel forward (6 Conv3d+ReLU on the depth-enabled GEMM, Linear, fused affine-grid "
                        "trilinear sampler) + Affloss, eager, 1 GPU"}
            log(f"3-D affmodel eval: {8 / t3:.1f} volumes/s")
            aff.train()
            opt_a = mireg.Adam(aff.parameters(), 1e-4, eps=1e-4)

            def aff_train():
                para, wv = aff(vol)
                loss = mireg.Affloss(wv, vol[:, 0:1])[2]
                opt_a.zero_grad()
                loss.backward()
                opt_a.step()
                return loss
            t3, l3 = timed(aff_train, 5)
            vol_line["affmodel_train"] = {"volumes_per_s": round(8 / t3, 1), "ms_per_step": round(t3 * 1e3, 3), "loss": float(l3.detach()),
                                          "note": "forward + Affloss + HIP backward (Conv3d backward-data / backward-weights, sampler "
                                                  "d/d theta) + Adam, eager"}
            log(f"3-D affmodel train: {8 / t3:.1f} volumes/s")
            del aff, opt_a
            torch.cuda.empty_cache()
            # BASELINE configs[4]: FlowNetS over 128^3 volumes, batch 8 -- full widths, train step
            reg3 = mireg.opticalFlowReg3d(precision=args.precision).to(dev).train()
            opt_f = mireg.Adam(reg3.parameters(), 1e-4, eps=1e-4, fuse=reg3)       # convolution weights: packed-domain Adam (mireg_adam_pack)

            def f3_train():
                flows, warped = reg3(vol)
                loss = mireg.OFEloss3d(flows, warped, vol[:, 0:1])[3]
                opt_f.zero_grad()
                loss.backward()
                opt_f.step()
                return loss.detach()      # a live loss keeps last step's AccumulateGrad nodes (and their stream) alive, which breaks capture
            t3, l3 = timed(f3_train, 6, warm=3)
            PROFILER.enabled, PROFILER.records, PROFILER.byte_records = True, [], []
            f3_train()
            s3 = PROFILER.summary()
            PROFILER.enabled = False
            fl3 = sum(v["flops"] for v in s3.values())
            ms3 = sum(v["ms"] for v in s3.values())
            roof3 = {"bound": "mfma", "kernel": "conv3d contractions (forward, backward-data, backward-weights)", "peak": PEAK_TFLOPS[args.precision],
                     "unit": "TFLOP/s", "achieved": round(fl3 / (ms3 * 1e-3) / 1e12, 1), "frac": round(fl3 / (ms3 * 1e-3) / 1e12 / PEAK_TFLOPS[args.precision], 4),
                     "gflop_per_step": round(fl3 / 1e9, 1), "contraction_ms_per_step": round(ms3, 3),
                     "step_tflops": round(fl3 / t3 / 1e12, 1), "traffic": None}
            vol_line["flownets3d_train"] = {"roofline": roof3,"volumes_per_s": round(8 / t3, 1), "ms_per_step": round(t3 * 1e3, 3), "loss": float(l3.detach()),
                                            "batch": 8, "note": "configs[4]: FlowNetS over 128^3 volume pairs (Conv3d / BatchNorm3d / "
                                            "ConvTranspose3d, 3-channel flow), six-scale warp + OFEloss3d, HIP backward, Adam; eager, 1 GPU"}
            log(f"3-D FlowNetS train: {8 / t3:.2f} volumes/s ({t3 * 1e3:.1f} ms/step)")
            # the same step replayed from one hipGraph (capturable since round 3: job tables are cached, gradient buffers persistent)
            try:
                import gc
                gr3, s3 = torch.cuda.CUDAGraph(), torch.cuda.Stream()
                s3.wait_stream(torch.cuda.current_stream())
                gc.disable()
                try:
                    with torch.cuda.stream(s3):
                        with torch.cuda.graph(gr3, stream=s3):
                            f3_train()
                finally:
                    gc.enable()
                tg, _ = timed(gr3.replay, 5, warm=1)
                vol_line["flownets3d_train"]["hipgraph"] = {"ms_per_step": round(tg * 1e3, 3), "volumes_per_s": round(8 / tg, 1)}
                log(f"3-D FlowNetS train, hipGraph replay: {tg * 1e3:.1f} ms/step")
                del gr3
            except Exception as e:                                   # noqa: BLE001
                vol_line["flownets3d_train"]["hipgraph"] = {"error": repr(e)}
            # quality on a genuine synthetic pair: moving = the fixed volume pushed through a smooth random displacement field
            # (on device), 4-label masks from intensity thresholds; warped-Dice after a short run from random init
            try:
                fixed3 = vol[:, 0:1].contiguous()
                fixed3 = (fixed3 - fixed3.amin()) / (fixed3.amax() - fixed3.amin())
                coarse = ((torch.rand(8, 3, 4, 4, 4, generator=g) * 2 - 1) * 5.0).to(dev)       # voxels
                field = mireg.resize_trilinear(coarse, (128, 128, 128), True)
                moving3 = mireg.stn3d(field, fixed3)
                pair = torch.cat((fixed3, moving3), 1).contiguous()
                segs3 = torch.bucketize(pair, torch.tensor([0.35, 0.5, 0.65], device=dev)).float()
                d_before = float(mireg.dice_batch(segs3[:, 0:1].contiguous(), segs3[:, 1:2].contiguous()).mean())
                for _ in range(30):
                    flows, warped = reg3(pair)
                    loss = mireg.OFEloss3d(flows, warped, pair[:, 0:1])[3]
                    opt_f.zero_grad()
                    loss.backward()
                    opt_f.step()
                reg3.eval()
                with torch.no_grad():
                    _, _, wseg3 = reg3(pair, segs3)
                d_after = float(mireg.dice_batch(segs3[:, 0:1].contiguous(), wseg3).mean())
                vol_line["flownets3d_dice"] = {"unregistered_mean": round(d_before, 5), "after_30_steps_mean": round(d_after, 5),
                                               "note": "synthetic 4-label volumes; 30 Adam steps from random init on one batch "
                                                       "(a plumbing check of the volume Dice chain, not a trained model)"}
            except Exception as e:                                   # noqa: BLE001
                vol_line["flownets3d_dice"] = {"error": repr(e)}
            del reg3, opt_f
            torch.cuda.empty_cache()
        except Exception as e:
            vol_line = dict(vol_line or {}, error=repr(e))

    # ---- BASELINE configs[2] / [3] on this GPU: FlowNetC (batch 24) and PWC-DC-Net (batch 48) train steps ----------------------
    others = None
    if rank == 0 and world == 1 and args.model == "flownets" and not args.no_other_models:
        others = {}
        del trainer, model
        torch.cuda.empty_cache()
        for nm, bsz in (("flownetc", 24), ("pwc", 48)):
            try:
                others[nm] = model_leg(nm, bsz, args.precision, dev)
                log(f"{nm} batch {bsz}: {others[nm]['pairs_per_s']} pairs/s ({others[nm]['ms_per_step']} ms/step)")
            except Exception as e:                                   # noqa: BLE001
                others[nm] = {"error": repr(e)}
        # the mode that meets north_star's "flow L2 < 1e-4": exact-fp32 MFMA operands (parity mode), same FlowNetS step, batch 24
        if not args.no_fp32_leg and args.precision == "bf16":
            try:
                others["flownets_fp32_parity_mode"] = model_leg("flownets", 24, "fp32", dev, steps=5, warm=3)
                others["flownets_fp32_parity_mode"]["note"] = ("configs[1] in fp32 parity mode (v_mfma_f32_32x32x2_f32 operands; flows match the "
                                                              "CPU oracle at 1e-4 + 4 x the reference's own fp32 noise, tests/test_bench_parity_gpu.py); "
                                                              "roofline peak = 157.3 TFLOP/s fp32 matrix")
                log(f"flownets fp32 batch 24: {others['flownets_fp32_parity_mode']['pairs_per_s']} pairs/s")
            except Exception as e:                                   # noqa: BLE001
                others["flownets_fp32_parity_mode"] = {"error": repr(e)}
        # the FlowNet2 stack, registration-wrapper shaped inference
        try:
            import mireg
            from mireg.synth import make_pairs
            reg2 = mireg.opticalFlowReg("flownet2", precision=args.precision).to(dev).eval()
            x2 = make_pairs(8, 256, seed=6)[0].to(dev)
            with torch.no_grad():
                for _ in range(2):
                    reg2(x2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    reg2(x2)
                torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / 5
            others["flownet2_eval"] = {"pairs_per_s": round(8 / dt2, 1), "ms_per_batch": round(dt2 * 1e3, 3), "batch": 8,
                                       "note": "opticalFlowReg('flownet2') forward + 2 warps, eval mode, eager, 162.5 M parameters"}
            log(f"flownet2 eval batch 8: {8 / dt2:.1f} pairs/s")
            # its training step as the reference runs it (train.py:48-57): forward, OFEloss, autograd backward (HIP per sub-network), Adam
            reg2.train()
            opt2 = mireg.Adam(reg2.parameters(), 1e-4, eps=1e-4, fuse=reg2)          # convolution weights: packed-domain Adam from the slabs

            def f2_train():
                flows, warped, _, _ = reg2(x2)
                loss = mireg.OFEloss(flows, warped, x2[:, 0:1])[3]
                opt2.zero_grad()
                loss.backward()
                opt2.step()
                return loss.detach()
            f2_train()

            def f2_fwd_bwd():                                        # launch shapes measured once on the real buffers (mireg.autotune)
                flows, warped, _, _ = reg2(x2)
                mireg.OFEloss(flows, warped, x2[:, 0:1])[3].backward()
            n_sites = mireg.autotune(reg2, f2_fwd_bwd)
            for _ in range(2):
                f2_train()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                l2 = f2_train()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / 4
            others["flownet2_train"] = {"pairs_per_s": round(8 / dt2, 1), "ms_per_step": round(dt2 * 1e3, 3), "batch": 8, "loss_total": float(l2.detach()),
                                        "note": "forward + OFEloss + HIP backward of the five sub-networks through torch.autograd + mireg.Adam(fuse=model), launch shapes from mireg.autotune, eager; hipgraph = the same step replayed from one hipGraph", "tuned_sites": n_sites}
            log(f"flownet2 train batch 8: {8 / dt2:.1f} pairs/s")
            try:                                                     # the same step replayed from one hipGraph (the eager form is host-bound)
                import gc
                gr2, s2 = torch.cuda.CUDAGraph(), torch.cuda.Stream()
                s2.wait_stream(torch.cuda.current_stream())
                gc.disable()
                try:
                    with torch.cuda.stream(s2):
                        with torch.cuda.graph(gr2, stream=s2):
                            f2_train()
                finally:
                    gc.enable()
                for _ in range(2):
                    gr2.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    gr2.replay()
                torch.cuda.synchronize()
                tg2 = (time.perf_counter() - t0) / 5
                others["flownet2_train"]["hipgraph"] = {"ms_per_step": round(tg2 * 1e3, 3), "pairs_per_s": round(8 / tg2, 1)}
                log(f"flownet2 train batch 8, hipGraph replay: {8 / tg2:.1f} pairs/s")
                del gr2
            except Exception as e:                                   # noqa: BLE001
                others["flownet2_train"]["hipgraph"] = {"error": repr(e)}
            del opt2
            del reg2
            torch.cuda.empty_cache()
        except Exception as e:                                       # noqa: BLE001
            others["flownet2_eval"] = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.batch, args.size, args.cpu_steps, seed=6)

    if rank == 0:
        out = {"metric": f"registration slice-pairs/s ({args.model} train step: fwd + warp + OFEloss + bwd + all-reduce + Adam)",
               "value": round(pairs / dt, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
               "config": {"workload": f"{ {'flownets': 'configs[1]', 'flownetc': 'configs[2] shape', 'pwc': 'configs[3] shape'}.get(args.model, 'custom') }: {args.model} {args.size}x{args.size} slice pairs, batch {args.batch}/GPU, "
                                      f"{args.precision} operands fp32 accumulate, train step", "global_batch": args.batch * world,
                          "parallelism": f"dp{world}", "hipgraph": not args.no_graph},
               "loss": {"photo": loss_vals[0], "corr": loss_vals[1], "smooth": loss_vals[2], "total": loss_vals[3]},
               "roofline": roof, "cpu_baseline": cpu, "dice": dice, "eval": ev_line, "volumes3d": vol_line, "other_models": others,
               "dist": {"world": world, "backend": (torch.distributed.get_backend() if world > 1 else None)}}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
