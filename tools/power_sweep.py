#!/usr/bin/env python3
"""What the chip sustains on the shipped lone-wave conv loop as a function of how many CUs run it (GPU box).

conv_row_lw_kernel on mod7's 3x3 1024 -> 2048, dilation 4, 8 x 128 x 256 (9.9 TFLOP per launch) with the persistent grid capped at
KDCC_PERSIST_CUS = 256 / 192 / 128 / 64 workgroups (= active CUs; the switch is read once per process, so every point is a fresh child):
  * shipped library, epilogue in:            TFLOP/s from HIP events; clock from a rocprofv3 --pmc GRBM_GUI_ACTIVE pass of the same child
  * tuning library, KDCC_CONV_TUNE=1024:     in-kernel clock = s_memtime ticks per 10-ns s_memrealtime tick, median over the workgroups
  * ... | 2048:                              the same with an L2-resident input (every tile stages image 0's rows 0 / d / 2d)
  * ... | 64:                                the main loop alone (no epilogue): the number DESIGN.md section 5 round 4 quoted from prose
usage: python tools/power_sweep.py [out.json]        (the parent never touches the GPU)"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, H, W, CIN, COUT, DIL = 8, 128, 256, 1024, 2048, 4
FLOP = 2.0 * N * H * W * COUT * 9 * CIN


def child():
    import ctypes as C
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import kdcc_amd
    if os.environ.get("KDCC_LIB") == "tuning":
        kdcc_amd._lib.build_tuning()
    from kdcc_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, H, W, CIN, device="cuda", generator=g).relu().bfloat16()
    w = (torch.randn(COUT, 3, 3, CIN, device="cuda", generator=g) * (2.0 / (9 * CIN)) ** 0.5).bfloat16()
    out = torch.empty(N, H, W, COUT, device="cuda", dtype=torch.bfloat16)
    sc, sh = torch.ones(COUT, device="cuda"), torch.zeros(COUT, device="cuda")
    run = lambda: ops.conv2d(x, w, 1, DIL, DIL, out_act=out, act_scale=sc, act_shift=sh, act_relu=True)
    with _lib.kernel_log() as log:
        run()
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 10)
    res = {"kernel": [k for k, v in log.counts.items() if v], "ms": min(ms), "tflops": FLOP / (min(ms) * 1e-3) / 1e12}
    tune = int(os.environ.get("KDCC_CONV_TUNE", "0"))
    if os.environ.get("KDCC_LIB") == "tuning" and tune & 1024:
        buf = np.zeros(512 * 32 * 8, dtype=np.uint64)
        lib = _lib.lib()
        lib.kd_debug_conv_tlog.argtypes = [C.c_void_p, C.c_size_t]
        lib.kd_debug_conv_tlog(buf.ctypes.data, buf.nbytes)
        t = buf.reshape(-1, 8)[:512].astype(np.int64)
        ok = (t[:, 3] > t[:, 1]) & (t[:, 2] > t[:, 0])
        ghz = (t[ok, 2] - t[ok, 0]) / ((t[ok, 3] - t[ok, 1]) * 10.0)        # shader ticks per ns
        res.update(workgroups_stamped=int(ok.sum()), clock_ghz_in_kernel=float(np.median(ghz)), clock_ghz_min=float(ghz.min()),
                   clock_ghz_max=float(ghz.max()), kernel_us_in_kernel=float(np.median((t[ok, 3] - t[ok, 1]) / 100.0)))
    print("RESULT " + json.dumps(res), flush=True)


def spawn(env_extra, rocprof_dir=None):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, os.path.abspath(__file__), "--child"]
    if rocprof_dir:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "-d", rocprof_dir, "-o", "p", "--output-format", "csv",
               "--"] + cmd
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if r.returncode != 0 or not line:
        raise SystemExit(f"child {env_extra} failed rc={r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}")
    res = json.loads(line[0][7:])
    if rocprof_dir:
        acc = {}
        for f in glob.glob(os.path.join(rocprof_dir, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "conv_row_lw_kernel" in row["Kernel_Name"]:
                    a = acc.setdefault(row["Counter_Name"], [0.0, 0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        a[1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                        a[2] += 1
        if "GRBM_GUI_ACTIVE" in acc and acc["GRBM_GUI_ACTIVE"][1] > 0:
            cyc, ns, n = acc["GRBM_GUI_ACTIVE"][0] / 8.0, acc["GRBM_GUI_ACTIVE"][1], acc["GRBM_GUI_ACTIVE"][2]
            res.update(clock_ghz_from_grbm=cyc / ns, launches_profiled=n, avg_us_under_rocprof=ns / n / 1e3,
                       tflops_under_rocprof=FLOP * n / (ns * 1e-9) / 1e12)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
                res["mfma_busy"] = acc["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (cyc * 1024.0)
    return res


def main():
    out = {"layer": f"3x3 dil {DIL} {CIN}->{COUT}, {N}x{H}x{W}, bf16, out_act (BN + ReLU epilogue), {FLOP / 1e12:.2f} TFLOP per launch",
           "peak_tflops_per_cu_at_2p4_ghz": 2500.0 / 256, "points": []}
    for cus in (256, 192, 128, 64):
        pt = {"workgroups": cus}
        base = {"KDCC_PERSIST_CUS": str(cus)}
        pt["shipped"] = spawn(base)
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
            pt["shipped_rocprof"] = spawn(base, rocprof_dir=d)
        pt["stamped"] = spawn(dict(base, KDCC_LIB="tuning", KDCC_CONV_TUNE="1024"))
        pt["stamped_l2_resident_input"] = spawn(dict(base, KDCC_LIB="tuning", KDCC_CONV_TUNE=str(1024 | 2048)))
        pt["stamped_no_epilogue"] = spawn(dict(base, KDCC_LIB="tuning", KDCC_CONV_TUNE=str(1024 | 64)))
        for k in ("shipped", "stamped", "stamped_l2_resident_input", "stamped_no_epilogue"):
            pt[k]["tflops_per_cu"] = pt[k]["tflops"] / cus
        out["points"].append(pt)
        print(json.dumps(pt), flush=True)
    p256 = out["points"][0]
    a, b = p256["stamped"], p256["stamped_l2_resident_input"]
    out["l2_resident_input_effect_at_256"] = {"clock_ratio": b["clock_ghz_in_kernel"] / a["clock_ghz_in_kernel"], "tflops_ratio": b["tflops"] / a["tflops"]}
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
