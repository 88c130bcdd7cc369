#!/usr/bin/env python3
"""Headline benchmark: images/sec of the KD train step, DeepLabV3+(WRN-38) student, 1024x2048 synthetic
Cityscapes-shaped tensors (BASELINE.json).  One process per GPU:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = what trainer/layerwise_trainer.py:220-239 of the reference does per batch: teacher forward (frozen, PyTorch-
ROCm), student forward (HIP engine), the four criteria (CE x2, KD, hint), loss = hint loss, backward, (N>1: bucketed
RCCL all-reduce of the trainable gradients, overlapped with backward), RAdam step, zero_grad.  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PLANS = {
    # P92 = 92.13 M-parameter student (BASELINE's "92M"); P79 = cfg/cityscapes/58M_deeplab_all.json verbatim
    "P92": ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod7.block1.convs.conv2",
            "aspp.features.1.0", "aspp.features.2.0", "aspp.features.3.0"],
    "P79": ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod4.block3.convs.conv2", "mod4.block4.convs.conv2",
            "mod4.block5.convs.conv2", "mod4.block6.convs.conv2", "mod5.block2.convs.conv2", "mod7.block1.convs.conv2",
            "aspp.features.1.0", "aspp.features.2.0", "aspp.features.3.0"],
}
PLANS["P86"] = ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod4.block3.convs.conv2", "mod4.block4.convs.conv2",
                "mod4.block6.convs.conv2", "mod7.block1.convs.conv2", "aspp.features.1.0", "aspp.features.2.0",
                "aspp.features.3.0"]   # cfg/cityscapes/51M_gscnn_all.json verbatim (the README's 86 M GSCNN student)
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md chip-level table
PEAK_F32_TFLOPS = 157.3


def build(plan, dtype, device, seed=123, mode="A", arch="deeplab"):
    import kdcc_amd
    from kdcc_amd import losses
    from kdcc_amd.models import GSCNN, DeepWV3Plus
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.utils.optim import RAdam
    torch.manual_seed(seed)                       # train.py:17-21 of the reference
    # random init: the Cityscapes checkpoints are not shipped
    teacher = (GSCNN(num_classes=19) if arch == "gscnn" else DeepWV3Plus(num_classes=19)).eval()
    cpu_sd = {k: v.detach().clone() for k, v in teacher.state_dict().items() if not k.endswith("num_batches_tracked")}
    model = DepthwiseStudent(teacher, None, dtype=dtype).to(device)
    model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
    model.register_hint_layers(plan)
    model.unfreeze(plan)
    if mode == "B":   # SURVEY 8(d) mode B: every student parameter trainable (dense convs, eval-mode BN affine, stem)
        for p in model.student.parameters():
            p.requires_grad = True
    crit = [losses.CrossEntropyLoss2d(ignore_index=255), losses.KLDivergenceLoss(1), losses.MSELoss(num_classes=1000)]
    opt = RAdam([p for p in model.student.parameters() if p.requires_grad], lr=0.005)
    return model, crit, opt, cpu_sd


def kd_step(model, crit, opt, data, target, mode="A"):
    out_st, out_tc = model(data)
    sup = crit[0](out_st, target)
    kd = crit[1](out_st, out_tc)
    tl = crit[0](out_tc, target)
    hint = 0
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        hint = hint + crit[2](s, t)
    loss = hint                      # "Only use hint loss", layerwise_trainer.py:233-235
    if mode == "B":
        loss = kd + hint             # north-star mode: the KD term is back-propagated too (classification_trainer.py:37)
    loss.backward()
    opt.step()
    opt.zero_grad()
    return loss, sup, kd, tl


def cpu_share():
    """Host cores this process can actually use: the affinity mask, cut to the cgroup CPU quota when there is one.  A GPU
    box exposes all 256 host cores in the mask but schedules a one-GPU job on a 16-core share; OpenMP threads beyond the
    share only spin (a 256-thread run of the CPU step did not finish in 7 minutes), so without a readable quota a mask
    wider than 32 cores is taken to be that case and 16 is used."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = max(1, q // per)
        except (OSError, ValueError):
            pass
    if quota is not None:
        return max(1, min(n, quota))
    return n if n <= 32 else 16


def cpu_baseline(cpu_sd, model, plan, full=False):
    """The network-level oracle (stock torch CPU ops, fp32) timed on this box's host cores.

    Default: a BOUNDED sample -- one 512x1024 step (a quarter of the pixels, ~15-30 s), scaled by pixel count -- so that the
    default bench run finishes in minutes.  full=True (--cpu-baseline full) is BASELINE.md section 3's recipe: 1 warm-up +
    2 timed steps at 1024x2048 (~1-2 min each) plus 5 timed steps at 256x512 for the linear-in-pixels check."""
    from oracle import net_ref
    threads = cpu_share()
    torch.set_num_threads(threads)
    hw = (1024, 2048) if full else (512, 1024)
    print(f"[bench] cpu_baseline: oracle/net_ref.py on {threads} host threads (os.cpu_count() = {os.cpu_count()}), "
          f"{'1 warm-up + 2 timed' if full else 'one'} {hw[0]}x{hw[1]} step(s) ...", file=sys.stderr, flush=True)
    new = {}
    for n in plan:
        blk = model.get_block(n, model.student)
        new[f"{n}.separable_conv.weight"] = blk.separable_conv.weight.detach().float().cpu()
        new[f"{n}.pointwise_conv.weight"] = blk.pointwise_conv.weight.detach().float().cpu()
    ssd = net_ref.make_student_sd(cpu_sd, plan, new)
    g = torch.Generator().manual_seed(1000)
    x = torch.randn((1, 3) + hw, generator=g)
    tgt = torch.randint(0, 19, (1,) + hw, generator=g)
    def timed(xx, tt, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            net_ref.kd_step(cpu_sd, ssd, xx, tt, plan)
            ts.append(time.perf_counter() - t0)
            print(f"[bench] cpu_baseline: {tuple(xx.shape[2:])} step {ts[-1]:.2f} s", file=sys.stderr, flush=True)
        return ts

    if full:
        timed(x, tgt, 1)                                   # warm-up at full size
        ts = timed(x, tgt, 2)
        small = timed(x[:, :, :256, :512].contiguous(), tgt[:, :256, :512].contiguous(), 5)
        dt = sum(ts) / len(ts)
        return {"value": 1.0 / dt, "unit": "images/sec", "cores": threads, "host_cpu_count": os.cpu_count(), "kind": "port",
                "sample": f"BASELINE.md section 3: 1 warm-up + 2 timed KD steps (teacher fwd + student fwd + hint bwd, fp32 torch "
                          f"CPU ops = oracle/net_ref.py) at 1024x2048, {ts[0]:.1f} / {ts[1]:.1f} s; 5 steps at 256x512: "
                          f"{sum(small) / len(small):.2f} s mean = {16 * sum(small) / len(small):.1f} s per full-size image equivalent",
                "steps_s": ts, "steps_256x512_s": small}
    net_ref.kd_step(cpu_sd, ssd, x[:, :, :64, :128].contiguous(), tgt[:, :64, :128].contiguous(), plan)  # warm-up
    dt = timed(x, tgt, 1)[0]
    frac = (hw[0] * hw[1]) / (1024.0 * 2048.0)
    return {"value": frac / dt, "unit": "images/sec", "cores": threads, "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"bounded sample (the bench must finish in minutes; --cpu-baseline full runs BASELINE.md section 3's 1+2 "
                      f"full-size steps): 1 KD step (teacher fwd + student fwd + hint bwd, fp32 torch CPU ops = "
                      f"oracle/net_ref.py) at {hw[0]}x{hw[1]} = {frac:.4f} of a 1024x2048 image, {dt:.2f} s; value = that "
                      f"fraction / time (measured linear in pixels from this size up: 54.8 s extrapolated vs 56-58 s per full-size step, DESIGN.md section 5)"}


def conv_traffic(plan, batch, height, width, dtype):
    """HBM bytes per conv_igemm launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate
    passes, same command); None when the profiled configuration is not the one being run."""
    path = os.path.join(ROOT, "profiles", "r02_traffic_pmc.json")
    if not (plan == "P92" and batch == 4 and (height, width) == (1024, 2048) and dtype == "bf16" and os.path.exists(path)):
        return None
    try:
        with open(path) as f:
            return float(json.load(f)["kernels"]["conv_igemm"]["hbm_bytes_per_launch"])
    except (KeyError, ValueError, OSError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--plan", default="P92", choices=sorted(PLANS))
    ap.add_argument("--arch", default="deeplab", choices=["deeplab", "gscnn"],
                    help="deeplab: DeepLabV3+(WRN-38), the headline (BASELINE configs 2-4); gscnn: Gated-SCNN teacher/student "
                         "(BASELINE config 5; use --plan P86, the shipped 51M_gscnn_all.json plan; mode A only)")
    ap.add_argument("--mode", default="A", choices=["A", "B"],
                    help="A (default, reference-faithful): loss = hint loss, only the cheap-conv blocks train; B (SURVEY 8d "
                         "north-star mode): loss = KLDiv + hint, every student parameter trainable (37.74 TFLOP/img for P92)")
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (4 by default: +6 %% img/s over 1, +2 %% over 2 from fuller grids)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full"],
                    help="sample: one 512x1024 CPU step scaled by pixels (default, bounded); full: BASELINE.md section 3 "
                         "(1 warm-up + 2 steps at 1024x2048 + 5 steps at 256x512; several minutes)")
    ap.add_argument("--layer-table", default=None, help="write a per-conv-shape timing table (tsv) to this path")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the 1 and 2 images/GPU side measurements")
    ap.add_argument("--no-overlap", action="store_true", help="with --teacher torch: run the teacher on the main stream")
    ap.add_argument("--ref-logging", action="store_true",
                    help="also do the reference's per-step host syncs (five .item() calls, layerwise_trainer.py:244-250); the "
                         "default measures the step without them, as this trainer runs it (metrics stay on the device)")
    ap.add_argument("--teacher-stream", default="main", choices=["main", "side"],
                    help="with --teacher hip: run the engine teacher on a side HIP stream concurrently with the student forward")
    ap.add_argument("--share-prefix", action="store_true",
                    help="opt-in: compute the frozen layers the student shares bit for bit with the teacher once per step "
                         "(stem .. the block before the first cheap conv); same numbers, ~8 %% fewer FLOPs than the reference's "
                         "two full forwards -- NOT the headline configuration")
    ap.add_argument("--teacher", default="hip", choices=["torch", "hip"],
                    help="hip: frozen teacher graph through the engine's HIP kernels (default); torch: teacher as a PyTorch-ROCm "
                         "module (MIOpen) on a side stream, the split north_star describes")
    a = ap.parse_args()

    import kdcc_amd
    from kdcc_amd import ops, parallel
    rank, local, world = parallel.init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    device = torch.device("cuda", local)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    plan = PLANS[a.plan]
    model, crit, opt, cpu_sd = build(plan, dtype, device, mode=a.mode, arch=a.arch)
    model.overlap_teacher = not a.no_overlap
    model.teacher_backend = a.teacher
    model.hip_teacher_side_stream = a.teacher_stream == "side"
    model.share_frozen_prefix = bool(a.share_prefix)
    if world > 1:
        eng = model._student_engine()
        eng.reducer = parallel.GradReducer(eng.grad_production_order())

    g = torch.Generator().manual_seed(1000 + rank)
    data = torch.randn((a.batch, 3, a.height, a.width), generator=g).to(device)
    target = torch.randint(0, 19, (a.batch, a.height, a.width), generator=g)
    target[:, :32] = 255
    target = target.to(device)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        kd_step(model, crit, opt, data, target, a.mode)
    sync()
    prof = []
    ops.PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, sup, kd, tl = kd_step(model, crit, opt, data, target, a.mode)
        if a.ref_logging:
            _ = (loss.item(), sup.item(), kd.item(), loss.item(), tl.item())
    sync()
    dt = time.perf_counter() - t0
    ops.PROFILER = None
    if a.layer_table and rank == 0:
        # per conv shape: launches per step, mean duration, TFLOP/s (live HIP events, same records as the roofline)
        agg = {}
        for p in prof:
            if len(p) > 4:
                e = agg.setdefault(p[4], [0, 0.0, p[1]])
                e[0] += 1
                e[1] += p[2].elapsed_time(p[3])
        with open(a.layer_table, "w") as f:
            f.write("shape\tlaunches_per_step\tavg_ms\tTFLOP/s\tms_per_step\n")
            for k, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{k}\t{cnt / a.steps:g}\t{ms / cnt:.4f}\t{fl * cnt / ms / 1e9:.0f}\t{ms / a.steps:.3f}\n")
    replicas_identical = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        # every rank trained on its own shard: the replicas stay bit-identical only if every gradient bucket was
        # all-reduced before the optimizer read it (fp64 sums of the trainable parameters, min == max over ranks)
        cs = torch.stack([p.detach().double().sum() for p in model.student.parameters() if p.requires_grad])
        lo, hi = cs.clone(), cs.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi)) and bool(torch.isfinite(cs).all())

    # SURVEY 8(d) defines the metric at 1 and 2 images per GPU; the headline uses --batch (fuller grids).  Measure both
    # side by side (short: 2 warm-up + 8 timed steps each) so one record carries all three.
    sweep = {}
    if not a.no_batch_sweep:
        for nb in (1, 2):
            if nb >= a.batch:
                continue
            d_, t_ = data[:nb].contiguous(), target[:nb].contiguous()
            for _ in range(2):
                kd_step(model, crit, opt, d_, t_, a.mode)
            sync()
            t1 = time.perf_counter()
            for _ in range(8):
                kd_step(model, crit, opt, d_, t_, a.mode)
            sync()
            d1 = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([d1], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                d1 = float(tt.item())
            sweep[str(nb)] = {"images_per_sec": world * nb * 8 / d1, "ms_per_step": d1 / 8 * 1e3, "steps": 8}

    # what actually ran: the PyTorch teacher overlaps on a side stream unless --no-overlap; the engine teacher runs in stream
    # order unless --teacher-stream side
    overlapped = bool(model.overlap_teacher) and (a.teacher == "torch" or a.teacher_stream == "side")
    if rank == 0:
        # dominant kernel: the implicit-GEMM conv; live HIP-event timing of every launch on its launching stream
        wg = [p for p in prof if p[0] == "conv_wgrad"]
        prof = [p for p in prof if p[0] != "conv_wgrad"]
        flops = sum(p[1] for p in prof)
        ms = sum(p[2].elapsed_time(p[3]) for p in prof)
        wg_flops, wg_ms = sum(p[1] for p in wg), sum(p[2].elapsed_time(p[3]) for p in wg)
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        res = {
            "metric": "images/sec KD train step, DeepLabV3+(WRN38) student 1024x2048",
            "value": world * a.batch * a.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"KD train step (frozen teacher fwd + student fwd + CE/KD/hint criteria + "
                                   f"{'hint-loss bwd into the cheap-conv blocks' if a.mode == 'A' else 'kd+hint bwd into every student parameter'} + "
                                   f"RAdam), DeepLabV3+(WRN-38) student plan {a.plan} ({len(plan)} cheap-conv blocks, 9x9 d5), "
                                   f"{a.height}x{a.width}, {a.batch} image/GPU, random-init weights",
                       "plan": a.plan, "mode": a.mode, "per_gpu_batch": a.batch, "global_batch": world * a.batch,
                       "parallelism": f"dp{world}", "replicas_identical_after_run": replicas_identical,
                       "teacher_overlap": overlapped,
                       "teacher_backend": a.teacher, "per_step_host_syncs": bool(a.ref_logging),
                       "share_frozen_prefix": bool(a.share_prefix),
                       "per_gpu_batch_sweep": sweep},
            "roofline": {"bound": "mfma", "kernel": "kd_conv2d_fwd: conv_row_persist_kernel + conv_igemm_persist_kernel + conv_row_pp128_kernel + conv_igemm_row_kernel + conv_igemm_kernel (dense conv fwd + dgrad, student" + (" + teacher)" if a.teacher == "hip" else ")"), "achieved": ach,
                         "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "traffic": conv_traffic(a.plan, a.batch, a.height, a.width, a.dtype) if a.mode == "A" else None,
                         "traffic_note": "mean HBM bytes per conv_igemm* launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + "
                                         "WRITE_SIZE, separate passes (profiles/r02_traffic_pmc.json); mean algorithmic "
                                         "FLOP per launch = algorithmic_tflop_per_step / launches_per_step",
                         "launches_per_step": len(prof) / max(a.steps, 1), "ms_per_step_in_kernel": ms / max(a.steps, 1),
                         "algorithmic_tflop_per_step": flops / max(a.steps, 1) / 1e12},
            "dense_wgrad": ({"kernel": "pw_wgrad_tr_kernel / pw_wgrad_kernel via kd_conv2d_wgrad", "achieved": wg_flops / (wg_ms * 1e-3) / 1e12,
                             "unit": "TFLOP/s", "launches_per_step": len(wg) / max(a.steps, 1), "ms_per_step_in_kernel": wg_ms / max(a.steps, 1),
                             "algorithmic_tflop_per_step": wg_flops / max(a.steps, 1) / 1e12} if wg_ms > 0 else None),
            "losses": {("hint" if a.mode == "A" else "kd+hint"): float(loss.detach()), "supervised": float(sup), "kd": float(kd), "teacher": float(tl)},
        }
        if a.arch == "gscnn":
            res["metric"] = "images/sec KD train step, Gated-SCNN (WRN38) student 1024x2048"
            res["config"]["workload"] = res["config"]["workload"].replace("DeepLabV3+(WRN-38) student", "Gated-SCNN (WRN-38 + shape stream, device Canny) student")
        if world == 1 and not a.no_cpu_baseline and a.arch == "deeplab":
            res["cpu_baseline"] = cpu_baseline(cpu_sd, model, plan, full=a.cpu_baseline == "full")
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
