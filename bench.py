#!/usr/bin/env python3
"""Headline benchmark: images/sec of the KD train step, DeepLabV3+(WRN-38) student, 1024x2048 synthetic
Cityscapes-shaped tensors (BASELINE.json).  One process per GPU:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W        (no launcher: bench.py starts that torch.distributed.run itself as a fresh
                                                          child process before anything touches a GPU, and relays rank 0's line)

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


def build(plan, dtype, device, seed=123, mode="A", arch="deeplab", hint_loss="mse"):
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
        model.logits_need_grad = True     # the KD term reaches the logits (Gated-SCNN: the shape stream keeps its intermediates)
    crit = [losses.CrossEntropyLoss2d(ignore_index=255), losses.KLDivergenceLoss(1),
            losses.WeightedHintMSELoss() if hint_loss == "weighted" else losses.MSELoss(num_classes=1000)]
    opt = RAdam([p for p in model.student.parameters() if p.requires_grad], lr=0.005)
    return model, crit, opt, cpu_sd


_FW = {}


def _filter_weight(i, channels, device):
    """BASELINE config 4 / SURVEY 8(d): filter_weight = rand(C) per hint, generator seed 7 + hint index (what
    trainer.hint_filter_weight = 'rand:7' gives LayerwiseTrainer)."""
    key = (i, channels, str(device))
    if key not in _FW:
        _FW[key] = torch.rand(channels, generator=torch.Generator().manual_seed(7 + i)).to(device)
    return _FW[key]


def kd_step(model, crit, opt, data, target, mode="A", prefetch=True):
    from kdcc_amd.losses import WeightedHintMSELoss
    out_st, out_tc = model(data)
    sup = crit[0](out_st, target)
    kd = crit[1](out_st, out_tc)
    tl = crit[0](out_tc, target)
    hint = 0
    weighted = isinstance(crit[2], WeightedHintMSELoss)
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        hint = hint + (crit[2](s, t, _filter_weight(i, s.shape[1], s.device)) if weighted else crit[2](s, t))
    loss = hint                      # "Only use hint loss", layerwise_trainer.py:233-235
    if mode == "B":
        loss = kd + hint             # north-star mode: the KD term is back-propagated too (classification_trainer.py:37)
    if prefetch and getattr(model, "prefetch_next", False):
        model.prefetch_teacher(data)     # --teacher-stream backward: the NEXT step's teacher forward (same synthetic batch) under this backward
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


def cpu_baseline(cpu_sd, model, plan, full="step"):
    """The network-level oracle (stock torch CPU ops, fp32) timed on this box's host cores.

    Default ("step"): a small warm-up, the bounded 512x1024 sample, then ONE REAL 1024x2048 KD step, timed; `value` is that
    measured full-size step (nothing is extrapolated), the 512x1024 sample is reported beside it.  "sample": the 512x1024
    step only (value = its pixel fraction / time; for quick local runs).  "full" (--cpu-baseline full): BASELINE.md
    section 3's recipe, 1 warm-up + 2 timed steps at 1024x2048 plus 5 timed steps at 256x512."""
    from oracle import net_ref
    threads = cpu_share()
    torch.set_num_threads(threads)
    print(f"[bench] cpu_baseline ({full}): oracle/net_ref.py on {threads} host threads (os.cpu_count() = {os.cpu_count()}) ...",
          file=sys.stderr, flush=True)
    new = {}
    for n in plan:
        blk = model.get_block(n, model.student)
        new[f"{n}.separable_conv.weight"] = blk.separable_conv.weight.detach().float().cpu()
        new[f"{n}.pointwise_conv.weight"] = blk.pointwise_conv.weight.detach().float().cpu()
    ssd = net_ref.make_student_sd(cpu_sd, plan, new)
    g = torch.Generator().manual_seed(1000)
    x = torch.randn((1, 3, 1024, 2048), generator=g)
    tgt = torch.randint(0, 19, (1, 1024, 2048), generator=g)

    def timed(h, w, reps):
        xx, tt = x[:, :, :h, :w].contiguous(), tgt[:, :h, :w].contiguous()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            net_ref.kd_step(cpu_sd, ssd, xx, tt, plan)
            ts.append(time.perf_counter() - t0)
            print(f"[bench] cpu_baseline: {h}x{w} step {ts[-1]:.2f} s", file=sys.stderr, flush=True)
        return ts

    base = {"unit": "images/sec", "cores": threads, "host_cpu_count": os.cpu_count(), "kind": "port"}
    what = "KD step (teacher fwd + student fwd + hint bwd, fp32 torch CPU ops = oracle/net_ref.py)"
    if full == "full":
        timed(1024, 2048, 1)                               # warm-up at full size
        ts = timed(1024, 2048, 2)
        small = timed(256, 512, 5)
        dt = sum(ts) / len(ts)
        return dict(base, value=1.0 / dt, steps_s=ts, steps_256x512_s=small,
                    sample=f"BASELINE.md section 3: 1 warm-up + 2 timed {what} at 1024x2048, {ts[0]:.1f} / {ts[1]:.1f} s; 5 steps at "
                           f"256x512: {sum(small) / len(small):.2f} s mean")
    timed(64, 128, 1)                                      # warm-up (thread pool, allocator)
    half = timed(512, 1024, 1)[0]
    if full == "sample":
        return dict(base, value=0.25 / half,
                    sample=f"bounded sample: 1 {what} at 512x1024 = a quarter of a 1024x2048 image, {half:.2f} s; value = 0.25 / time")
    one = timed(1024, 2048, 1)[0]
    return dict(base, value=1.0 / one, step_1024x2048_s=one, step_512x1024_s=half,
                sample=f"one measured {what} at the full 1024x2048, 1 image: {one:.1f} s (after a 64x128 warm-up and one 512x1024 "
                       f"step, {half:.2f} s = {0.25 / half:.4f} img/s pixel-scaled, reported beside it); value = 1 / that step, "
                       f"nothing extrapolated")


def _latest_profile(suffix):
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{suffix}")))
    return c[-1] if c else None


def conv_traffic(plan, batch, height, width, dtype, mode, arch):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE in separate passes,
    same command); None when the profiled configuration is not the one being run."""
    want = {("A", "deeplab", "P92"): "traffic_pmc.json", ("B", "deeplab", "P92"): "traffic_modeB_pmc.json",
            ("A", "gscnn", "P86"): "traffic_gscnn_pmc.json"}.get((mode, arch, plan))
    path = _latest_profile(want) if want else None
    if not ((height, width) == (1024, 2048) and dtype == "bf16" and path):
        return None, None
    try:
        with open(path) as f:
            d = json.load(f)
        if int(d.get("config", {}).get("batch", 4)) != batch:
            return None, None
        return float(d["kernels"]["conv_igemm"]["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
    except (KeyError, ValueError, OSError):
        return None, None


PEAK_HBM_GBS = 8000.0       # HBM3E, MI355X_MICROARCH.md

# device kernel (kernel-selection log) -> roofline class
CONV_CLASSES = (("conv_row_lw_kernel", "conv3x3_row_lone_wave_256x256"), ("conv_row_persist_kernel", "conv3x3_row_persistent_256x256"), ("conv_igemm_persist_kernel", "conv1x1_persistent_256x256"),
                ("conv_row_tall_kernel", "conv3x3_row_lone_wave_512x128"), ("conv_row_pp128_kernel", "conv3x3_row_512x128"))


# depthwise classes with a floor record: (tensors one launch moves, MFMAs per (output element, branch)): a 13 x 52 item-channel-branch is 4 column tiles x 7
# MFMAs in the forward / input-gradient kernels (K slots), 21 rows x 2 column blocks of Hankel MFMAs in the weight gradient
DW_FLOORS = {"depthwise_fwd_fanout": (4, 28 / (13 * 52)), "depthwise_dgrad_sum": (4, 28 / (13 * 52)), "depthwise_wgrad": (4, 42 / (13 * 52))}
DW_CLOCK_HZ = 2.1e9


def _classify(rec):
    fam, kern, label = rec[0], (rec[5] if len(rec) > 5 else ""), rec[4]
    if fam == "conv_igemm":
        for prefix, name in CONV_CLASSES:
            if kern.startswith(prefix):
                return name
        return "conv_other_tiles"
    if fam == "conv_wgrad":
        return "dense_wgrad"
    if fam == "pw_wgrad":
        return "pointwise_wgrad"
    if fam == "depthwise":
        if "wgrad" in label:
            return "depthwise_wgrad"
        if "fan-out" in label:
            return "depthwise_fwd_fanout"
        if "sum of" in label:
            return "depthwise_dgrad_sum"
        return "depthwise_dgrad_epilogue" if "+epi" in label else "depthwise_fwd"
    if fam == "channel_sums":
        return "bn_param_sums"
    return "losses"


def class_rooflines(prof, steps, peak_tflops):
    """Per kernel class: live HIP-event time per step, launches per step, achieved rate and fraction of the roof that bounds
    it (MFMA classes: algorithmic FLOP / time / dense bf16 peak; HBM classes: algorithmic bytes / time / 8 TB/s)."""
    agg = {}
    for rec in prof:
        e = agg.setdefault(_classify(rec), [0, 0.0, 0.0, rec[0]])
        e[0] += 1
        e[1] += rec[2].elapsed_time(rec[3])
        e[2] += rec[1]
    out = {}
    for name, (cnt, ms, work, fam) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if ms <= 0:
            continue
        hbm = fam in ("depthwise", "loss", "channel_sums")
        rate = work / (ms * 1e-3) / (1e9 if hbm else 1e12)
        out[name] = {"bound": "hbm" if hbm else "mfma", "launches_per_step": cnt / steps, "ms_per_step": ms / steps,
                     "achieved": rate, "unit": "GB/s" if hbm else "TFLOP/s", "peak": PEAK_HBM_GBS if hbm else peak_tflops,
                     "frac": rate / (PEAK_HBM_GBS if hbm else peak_tflops),
                     ("algorithmic_gb_per_step" if hbm else "algorithmic_tflop_per_step"): work / steps / (1e9 if hbm else 1e12)}
        if name in DW_FLOORS:
            # the class's true floors: matrix work at 16 cycles per MFMA on 1024 SIMDs, bytes at the 6.3 TB/s a
            # contiguous stream gets, and bytes at the 2.5 TB/s the memory system delivers on these kernels' 32-B channel slices
            # (profiles/r06_dw_anatomy.md) -- the HBM fraction above is against 8 TB/s
            tensors, mfma_per_elem = DW_FLOORS[name]
            elems = work / steps / 2.0 / tensors * (tensors - 1)          # (output element, branch) pairs per step
            out[name]["floor_ms"] = {"matrix": elems * mfma_per_elem * 16 / (1024 * DW_CLOCK_HZ) * 1e3, "bytes_at_6.3TBps": work / steps / 6.3e12 * 1e3,
                                     "bytes_at_2.5TBps_pattern": work / steps / 2.5e12 * 1e3}
    return out


def run_config(a, device, rank, world, plan_name=None, mode=None, arch=None, hint_loss="mse", steps=None, warmup=None,
               batch_sweep=True, dtype_name=None, batch=None, share_prefix=None, ref_logging=None):
    """Build one configuration, run `warmup` untimed + `steps` timed KD train steps, return (record, model, cpu_sd, plan)."""
    from kdcc_amd import ops, parallel
    plan_name, mode, arch = plan_name or a.plan, mode or a.mode, arch or a.arch
    steps, warmup = steps or a.steps, a.warmup if warmup is None else warmup
    if dtype_name is not None or batch is not None:       # a sub-record in another storage type / batch (the fp32 parity path)
        a = argparse.Namespace(**{**vars(a), "dtype": dtype_name or a.dtype, "batch": batch or a.batch})
    if share_prefix is not None or ref_logging is not None:      # sub-records: the opt-in shared frozen prefix / the reference's per-step host syncs
        a = argparse.Namespace(**{**vars(a), "share_prefix": a.share_prefix if share_prefix is None else share_prefix,
                                  "ref_logging": a.ref_logging if ref_logging is None else ref_logging})
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    plan = PLANS[plan_name]
    model, crit, opt, cpu_sd = build(plan, dtype, device, mode=mode, arch=arch, hint_loss=hint_loss)
    model.overlap_teacher = not a.no_overlap
    model.teacher_backend = a.teacher
    model.hip_teacher_side_stream = a.teacher_stream == "side"
    model.prefetch_next = a.teacher_stream == "backward"
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

    for _ in range(warmup):
        kd_step(model, crit, opt, data, target, mode)
    sync()
    prof = []
    ops.PROFILER = None if a.no_profiler else prof      # two HIP events per kernel launch inside the timed region (--no-profiler: none)
    t0 = time.perf_counter()
    for _ in range(steps):
        # (--teacher-stream backward: every step prefetches the next one's teacher, the last one included; the window is closed by
        # torch.cuda.synchronize(), a DEVICE-wide wait, so that last forward is finished inside it: the first step's teacher ran
        # during warm-up, the last step's extra one replaces it -- exactly `steps` teacher forwards are timed)
        loss, sup, kd, tl = kd_step(model, crit, opt, data, target, mode)
        if a.ref_logging:
            _ = (loss.item(), sup.item(), kd.item(), loss.item(), tl.item())
    sync()
    dt = time.perf_counter() - t0
    ops.PROFILER = None
    # what the per-launch events cost: the same step, un-instrumented, right behind the timed region (same device, same clocks)
    ab_steps = 0 if (a.no_profiler or a.no_profiler_ab or not batch_sweep) else min(8, steps)
    dt_plain = None
    if ab_steps:
        t1 = time.perf_counter()
        for _ in range(ab_steps):
            kd_step(model, crit, opt, data, target, mode)
        sync()
        dt_plain = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt_plain], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt_plain = float(tt.item())
    if a.layer_table and rank == 0 and batch_sweep:
        # per conv shape: launches per step, mean duration, TFLOP/s (live HIP events, same records as the roofline)
        agg = {}
        for p in prof:
            if p[0] in ("conv_igemm", "conv_wgrad", "pw_wgrad", "depthwise", "channel_sums"):
                e = agg.setdefault((p[4], p[5]), [0, 0.0, p[1], p[0]])
                e[0] += 1
                e[1] += p[2].elapsed_time(p[3])
        with open(a.layer_table, "w") as f:
            f.write("shape\tkernel\tlaunches_per_step\tavg_ms\tTFLOP/s|GB/s\tms_per_step\n")
            for (k, kern), (cnt, ms, work, fam) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{k}\t{kern}\t{cnt / steps:g}\t{ms / cnt:.4f}\t{work * cnt / ms / (1e6 if fam in ('depthwise', 'channel_sums') else 1e9):.0f}\t{ms / steps:.3f}\n")
    replicas_identical = None
    # what the process group actually was (not an echo of --gpus): ranks, backend, the device each rank ran on
    ranks_seen, backend = 1, None
    devices = [{"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(),
                "pci_bus_id": getattr(torch.cuda.get_device_properties(device), "pci_bus_id", None)}]
    if world > 1:
        ranks_seen, backend = torch.distributed.get_world_size(), torch.distributed.get_backend()
        box = [None] * ranks_seen
        torch.distributed.all_gather_object(box, devices[0])
        devices = box
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
    if batch_sweep and not a.no_batch_sweep:
        for nb in (1, 2, 4):
            if nb >= a.batch:
                continue
            d_, t_ = data[:nb].contiguous(), target[:nb].contiguous()
            for _ in range(2):
                kd_step(model, crit, opt, d_, t_, mode)
            sync()
            t1 = time.perf_counter()
            for _ in range(8):
                kd_step(model, crit, opt, d_, t_, mode)
            sync()
            d1 = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([d1], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                d1 = float(tt.item())
            sweep[str(nb)] = {"images_per_sec": world * nb * 8 / d1, "ms_per_step": d1 / 8 * 1e3, "steps": 8}

    # what actually ran: the PyTorch teacher overlaps on a side stream unless --no-overlap; the engine teacher runs in stream
    # order unless --teacher-stream side
    overlapped = (bool(model.overlap_teacher) and (a.teacher == "torch" or a.teacher_stream == "side")) or a.teacher_stream == "backward"
    res = None
    if rank == 0:
        # dominant kernel family: the implicit-GEMM conv (forward + input gradient); live HIP-event timing of every launch on
        # its launching stream
        conv = [p for p in prof if p[0] == "conv_igemm"]
        wg = [p for p in prof if p[0] == "conv_wgrad"]
        flops = sum(p[1] for p in conv)
        ms = sum(p[2].elapsed_time(p[3]) for p in conv)
        wg_flops, wg_ms = sum(p[1] for p in wg), sum(p[2].elapsed_time(p[3]) for p in wg)
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        traffic, traffic_src = conv_traffic(plan_name, a.batch, a.height, a.width, a.dtype, mode, arch)
        arch_name = "Gated-SCNN (WRN-38 + shape stream, device Canny)" if arch == "gscnn" else "DeepLabV3+(WRN-38)"
        crit_name = "WeightedHintMSELoss (filter_weight = rand(C), seed 7 + hint index)" if hint_loss == "weighted" else "MSELoss(num_classes=1000)"
        res = {
            "metric": ("images/sec KD train step, Gated-SCNN (WRN38) student 1024x2048" if arch == "gscnn"
                       else "images/sec KD train step, DeepLabV3+(WRN38) student 1024x2048"),
            "value": world * a.batch * steps / dt, "unit": "images/sec", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "profiler": {"events_in_timed_region": not a.no_profiler,
                         "ms_per_step_without_events": (dt_plain / ab_steps * 1e3 if ab_steps else None), "ab_steps": ab_steps},
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"KD train step (frozen teacher fwd + student fwd + CE/KD/hint criteria, hint = {crit_name}, + "
                                   f"{'hint-loss bwd into the cheap-conv blocks' if mode == 'A' else 'kd+hint bwd into every student parameter'} + "
                                   f"RAdam), frozen teacher through {'the HIP engine (teacher_backend hip), stream order' if a.teacher == 'hip' else 'PyTorch-ROCm on a side stream (teacher_backend torch)'}, "
                                   f"{arch_name} student plan {plan_name} ({len(plan)} cheap-conv blocks, 9x9 d5), "
                                   f"{a.height}x{a.width}, {a.batch} image/GPU, random-init weights",
                       "plan": plan_name, "mode": mode, "arch": arch, "hint_loss": hint_loss, "per_gpu_batch": a.batch,
                       "global_batch": world * a.batch, "parallelism": f"dp{world}",
                       "ranks_seen": ranks_seen, "backend": backend, "rank_devices": devices,
                       "replicas_identical_after_run": replicas_identical, "teacher_overlap": overlapped,
                       "teacher_backend": a.teacher, "per_step_host_syncs": bool(a.ref_logging),
                       "share_frozen_prefix": bool(a.share_prefix), "per_gpu_batch_sweep": sweep},
            "roofline": {"bound": "mfma",
                         "kernel": "kd_conv2d_fwd: conv_row_lw_kernel + conv_igemm_persist_kernel + conv_row_tall_kernel + conv_igemm_row_kernel "
                                   "+ conv_igemm_kernel (dense conv fwd + dgrad, student" + (" + teacher)" if a.teacher == "hip" else ")"),
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_src": traffic_src,
                         "traffic_note": ("mean HBM bytes per conv launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate "
                                          f"passes of this command ({traffic_src}); mean algorithmic FLOP per launch = "
                                          "algorithmic_tflop_per_step / launches_per_step") if traffic is not None else
                                         "no committed PMC pass for this configuration",
                         "launches_per_step": len(conv) / max(steps, 1), "ms_per_step_in_kernel": ms / max(steps, 1),
                         "algorithmic_tflop_per_step": flops / max(steps, 1) / 1e12,
                         "classes": class_rooflines(prof, max(steps, 1), peak)},
            "dense_wgrad": ({"kernel": "conv_wgrad_lw_kernel (3x3) / conv_wgrad_pw_lw_kernel (1x1) / conv_wgrad_wide_kernel / pw_wgrad_tr_kernel via kd_conv2d_wgrad",
                             "achieved": wg_flops / (wg_ms * 1e-3) / 1e12, "unit": "TFLOP/s", "peak": peak,
                             "frac": wg_flops / (wg_ms * 1e-3) / 1e12 / peak, "launches_per_step": len(wg) / max(steps, 1),
                             "ms_per_step_in_kernel": wg_ms / max(steps, 1),
                             "algorithmic_tflop_per_step": wg_flops / max(steps, 1) / 1e12} if wg_ms > 0 else None),
            "losses": {("hint" if mode == "A" else "kd+hint"): float(loss.detach()), "supervised": float(sup), "kd": float(kd), "teacher": float(tl)},
        }
    return res, model, cpu_sd, plan


def _r(x, nd=4):
    """floats to nd significant digits (the compact line is read by a driver that keeps only a few KB of stdout)"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    return x


def compact_record(res, full_path=None):
    """The ONE stdout line: contract keys + roofline + cpu_baseline + one number per side measurement, < 4 KB.  Everything
    else (per-class detail, layer notes, sub-record rooflines) is in the full record file / on stderr."""
    cfg, rf = res["config"], res["roofline"]
    out = {k: _r(res[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": (f"KD train step (teacher fwd + student fwd + CE/KD/hint criteria + "
                                  f"{'hint bwd' if cfg['mode'] == 'A' else 'kd+hint bwd, all params'} + RAdam), "
                                  f"{'Gated-SCNN' if cfg['arch'] == 'gscnn' else 'DeepLabV3+'}(WRN-38) student {cfg['plan']}, "
                                  f"{res.get('_hw', '1024x2048')}, {cfg['per_gpu_batch']} img/GPU, random init, teacher through the "
                                  f"{'HIP engine' if cfg['teacher_backend'] == 'hip' else 'PyTorch-ROCm side stream'}"),
                     **{k: cfg[k] for k in ("plan", "mode", "arch", "hint_loss", "per_gpu_batch", "global_batch", "parallelism",
                                            "teacher_backend", "share_frozen_prefix", "replicas_identical_after_run",
                                            "ranks_seen", "backend")},
                     "rank_devices": [d["device"] for d in cfg["rank_devices"]]}
    out["roofline"] = {"bound": rf["bound"], "kernel": "kd_conv2d_fwd (dense conv fwd + dgrad: conv_row_lw / conv_igemm_persist / "
                                                       "conv_row_tall / one-tile kernels)",
                       **{k: _r(rf.get(k)) for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_src", "launches_per_step",
                                                 "ms_per_step_in_kernel", "algorithmic_tflop_per_step")},
                       "classes": {n: [_r(c["ms_per_step"], 3), _r(c["frac"], 3)] for n, c in rf["classes"].items()},
                       "dw_floor_ms": {n: [_r(c["floor_ms"]["matrix"], 3), _r(c["floor_ms"]["bytes_at_6.3TBps"], 3), _r(c["floor_ms"]["bytes_at_2.5TBps_pattern"], 3)]
                                       for n, c in rf["classes"].items() if "floor_ms" in c},
                       "dw_floor_fmt": "[matrix, bytes / 6.3 TB/s, bytes / 2.5 TB/s (measured cap of 32-B channel slices)] ms per step",
                       "classes_fmt": "[ms_per_step, frac of its roof (mfma 2500 TFLOP/s | hbm 8000 GB/s)]"}
    if res.get("profiler"):
        out["profiler"] = {k: _r(v) for k, v in res["profiler"].items()}
    cb = res.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "host_cpu_count", "kind", "step_1024x2048_s",
                                                      "step_512x1024_s") if k in cb}
        out["cpu_baseline"]["sample"] = ("one measured 1024x2048 KD step of oracle/net_ref.py (fp32 torch CPU ops), 1 image"
                                         if "step_1024x2048_s" in cb else cb.get("sample", "")[:160])
    if res.get("sub_records"):
        out["sub_records"] = {n: {"value": _r(r["value"]), "ms_per_step": _r(r["ms_per_step"]), "dtype": r.get("dtype"), "frac": _r(r["roofline"]["frac"], 3),
                                  "traffic": _r(r["roofline"].get("traffic")),
                                  "wgrad_frac": _r(r["dense_wgrad"]["frac"], 3) if r.get("dense_wgrad") else None}
                              for n, r in res["sub_records"].items()}
    sw = cfg.get("per_gpu_batch_sweep") or {}
    if sw:
        out["batch_sweep"] = {n: _r(v["images_per_sec"]) for n, v in sw.items()}
    out["losses"] = {k: _r(v, 6) for k, v in res.get("losses", {}).items()}
    if full_path:
        out["full_record"] = full_path
    return out


# sub-records of the default run: the other BASELINE configurations, short (2 warm-up + 8 timed steps), same JSON line
SUB_RECORDS = (
    ("P79", dict(plan_name="P79"), "the shipped cfg/cityscapes/58M_deeplab_all.json plan (79.75 M-parameter student)"),
    ("modeB", dict(mode="B"), "north-star mode B: loss = KLDiv + hints, all 92.1 M student parameters trainable (37.74 TFLOP/img)"),
    ("gscnn_P86", dict(arch="gscnn", plan_name="P86"), "BASELINE config 5: Gated-SCNN student, cfg/cityscapes/51M_gscnn_all.json plan"),
    ("weighted_hint", dict(hint_loss="weighted"), "BASELINE config 4: WeightedHintMSELoss feature-hint KD, filter_weight = rand(C) (rand:7)"),
    ("share_prefix", dict(share_prefix=True),
     "opt-in DepthwiseStudent.share_frozen_prefix: the frozen layers the student shares bit for bit with the teacher (stem .. the block before the "
     "first cheap conv) computed once per step -- same numbers, ~8 % fewer FLOPs than the reference's two full forwards; NOT the headline"),
    ("ref_logging", dict(ref_logging=True),
     "SURVEY 8(d) 'with the reference's per-step logging syncs': the headline step plus the five .item() host syncs of "
     "trainer/layerwise_trainer.py:244-250 after every step"),
    ("f32_parity", dict(dtype_name="f32", batch=2, steps=2, warmup=1),
     "BASELINE.md section 4 'fp32 parity mode separately': the same P92 step with fp32 storage and fp32 MFMA (the path the 1e-3 parity tests run), "
     "2 images, 1 warm-up + 2 timed steps; frac against the 157.3 TFLOP/s fp32 matrix peak"),
)


def _launch_ranks(n):
    """Run this same command line as `n` ranks of one node (one process per GPU over RCCL; reference: `n_gpu` in the config is all
    a user sets, base/base_trainer.py:16-19).  The ranks are children of a `python -m torch.distributed.run` child; their stdout is
    read line by line: rank 0's JSON record goes to this process's stdout, anything else to stderr.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL fails with hipIpcGetMemHandle otherwise
    env.setdefault("OMP_NUM_THREADS", str(max(1, cpu_share() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {n} without WORLD_SIZE: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    for line in proc.stdout:
        is_record = False
        if line.startswith("{"):
            try:
                is_record = "metric" in json.loads(line)
            except ValueError:
                pass
        (sys.stdout if is_record else sys.stderr).write(line)
        (sys.stdout if is_record else sys.stderr).flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--plan", default="P92", choices=sorted(PLANS))
    ap.add_argument("--arch", default="deeplab", choices=["deeplab", "gscnn"],
                    help="deeplab: DeepLabV3+(WRN-38), the headline (BASELINE configs 2-4); gscnn: Gated-SCNN teacher/student "
                         "(BASELINE config 5; use --plan P86, the shipped 51M_gscnn_all.json plan)")
    ap.add_argument("--mode", default="A", choices=["A", "B"],
                    help="A (default, reference-faithful): loss = hint loss, only the cheap-conv blocks train; B (SURVEY 8d "
                         "north-star mode): loss = KLDiv + hint, every student parameter trainable (37.74 TFLOP/img for P92)")
    ap.add_argument("--hint-loss", default="mse", choices=["mse", "weighted"],
                    help="mse: MSELoss(num_classes=1000) (configs 2/3); weighted: WeightedHintMSELoss with rand:7 filter weights (config 4)")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (8 by default: fuller grids, +8 %% img/s over 1, +4 %% over 2, +2 %% over 4; the record carries 1, 2 and 4 beside it)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="step", choices=["step", "sample", "full"],
                    help="step (default): one measured 1024x2048 CPU step (~1 min) after a 512x1024 one; sample: the 512x1024 step only, "
                         "pixel-scaled; full: BASELINE.md section 3 (1 warm-up + 2 steps at 1024x2048 + 5 steps at 256x512)")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip the short P79 / mode B / GSCNN P86 / WeightedHintMSE side measurements of the default run")
    ap.add_argument("--layer-table", default=None, help="write a per-conv-shape timing table (tsv) to this path")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the 1 and 2 images/GPU side measurements")
    ap.add_argument("--no-profiler", action="store_true",
                    help="time the steps without the two HIP events per kernel launch (no roofline in the record); the default run "
                         "measures the same un-instrumented step right behind the timed region and reports both")
    ap.add_argument("--no-profiler-ab", action="store_true", help="skip that un-instrumented A/B measurement")
    ap.add_argument("--full-record", default=None,
                    help="where the full record goes (default gpurun_out/bench_full.json); stdout carries the compact line only")
    ap.add_argument("--no-overlap", action="store_true", help="with --teacher torch: run the teacher on the main stream")
    ap.add_argument("--ref-logging", action="store_true",
                    help="also do the reference's per-step host syncs (five .item() calls, layerwise_trainer.py:244-250); the "
                         "default measures the step without them, as this trainer runs it (metrics stay on the device)")
    ap.add_argument("--teacher-stream", default="main", choices=["main", "side", "backward"],
                    help="with --teacher hip: side = the engine teacher on a side HIP stream concurrently with the student forward; "
                         "backward = the teacher's forward for the next step launched on the side stream right before this step's "
                         "loss.backward() (DepthwiseStudent.prefetch_teacher: the placement north_star names)")
    ap.add_argument("--share-prefix", action="store_true",
                    help="opt-in: compute the frozen layers the student shares bit for bit with the teacher once per step "
                         "(stem .. the block before the first cheap conv); same numbers, ~8 %% fewer FLOPs than the reference's "
                         "two full forwards -- NOT the headline configuration")
    ap.add_argument("--teacher", default="hip", choices=["torch", "hip"],
                    help="hip: frozen teacher graph through the engine's HIP kernels (default); torch: teacher as a PyTorch-ROCm "
                         "module (MIOpen) on a side stream, the split north_star describes")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed without a launcher: start the N ranks as FRESH child processes.  This process has not
        # touched the GPU (no torch.cuda call, no kdcc_amd import) and never does; nothing is re-exec'd.
        raise SystemExit(_launch_ranks(a.gpus))

    import kdcc_amd
    from kdcc_amd import parallel
    rank, local, world = parallel.init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    device = torch.device("cuda", local)

    res, model, cpu_sd, plan = run_config(a, device, rank, world, hint_loss=a.hint_loss)
    headline_default = (a.plan, a.mode, a.arch, a.hint_loss, a.dtype, a.teacher, (a.height, a.width)) == \
        ("P92", "A", "deeplab", "mse", "bf16", "hip", (1024, 2048)) and not a.share_prefix
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.arch == "deeplab":
        res["cpu_baseline"] = cpu_baseline(cpu_sd, model, plan, full=a.cpu_baseline)
    del model
    if world == 1 and headline_default and not a.no_sub_records:
        import gc
        subs = {}
        for name, kw, what in SUB_RECORDS:
            gc.collect()
            torch.cuda.empty_cache()
            print(f"[bench] sub-record {name}: {what} ...", file=sys.stderr, flush=True)
            r, m, _, _ = run_config(a, device, rank, world, **{"steps": 8, "warmup": 2, "batch_sweep": False, **kw})
            del m
            keep = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "dense_wgrad", "losses")}
            keep["what"] = what
            keep["config"] = {k: r["config"][k] for k in ("plan", "mode", "arch", "hint_loss", "per_gpu_batch", "share_frozen_prefix", "per_step_host_syncs")}
            keep["roofline"]["peak_tflops"] = r["roofline"]["peak"]
            subs[name] = keep
        res["sub_records"] = subs
    if rank == 0:
        res["_hw"] = f"{a.height}x{a.width}"
        # the full record (per-class rooflines with their notes, sub-record rooflines, the batch sweep) goes to a file and to
        # stderr; stdout carries ONE compact line (the driver keeps a few KB of it)
        full = a.full_record or os.path.join(ROOT, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(full), exist_ok=True)
            with open(full, "w") as f:
                json.dump(res, f)
            shown = os.path.relpath(full, ROOT)
        except OSError:
            shown = None
        print("[bench] full record: " + json.dumps(res), file=sys.stderr, flush=True)
        line = json.dumps(compact_record(res, shown))
        assert len(line) < 4096, len(line)
        print(line, flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
