"""CPU-only: the C-ABI library loads and exports every symbol include/kdcc.h declares; no compute calls."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import kdcc_amd
    return kdcc_amd


def test_exports_every_declared_symbol(built):
    from kdcc_amd import _lib
    header = open(os.path.join(ROOT, "include", "kdcc.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(kd_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in kdcc.h but not exported"
    assert sorted(_lib.exported_symbols()) == declared, "ctypes binding table and kdcc.h disagree"
    assert _lib.lib().kd_version() >= 100


def test_no_cpu_fallback(built):
    """The product path refuses CPU tensors instead of silently computing somewhere else."""
    from kdcc_amd import losses, ops
    from kdcc_amd._lib import KdccError
    with pytest.raises(KdccError):
        ops.hint_mse(torch.zeros(1, 4, 2, 2), torch.zeros(1, 4, 2, 2), 1000)
    with pytest.raises(KdccError):
        losses.MSELoss(num_classes=1000)(torch.zeros(1, 4, 2, 2, requires_grad=True), torch.zeros(1, 4, 2, 2))


def test_nn_hip_refuses_host_tensors_unless_told(built):
    """nn_hip.Conv2d / BatchNorm2d (CIFAR config) raise on host tensors; only the explicit host plumbing mode (n_gpu: 0) may
    run torch's own ops."""
    from kdcc_amd import nn_hip
    from kdcc_amd._lib import KdccError
    conv, bn = nn_hip.Conv2d(3, 4, 3, padding=1), nn_hip.BatchNorm2d(4)
    x = torch.zeros(1, 3, 8, 8)
    nn_hip.allow_host_tensors(False)
    with pytest.raises(KdccError):
        conv(x)
    with pytest.raises(KdccError):
        bn(torch.zeros(1, 4, 8, 8))
    nn_hip.allow_host_tensors(True)
    try:
        assert tuple(bn(conv(x)).shape) == (1, 4, 8, 8)
    finally:
        nn_hip.allow_host_tensors(False)


def test_default_library_ignores_the_tuning_switches(built):
    """KDCC_CONV_TUNE / KDCC_DW_DBG / KDCC_WGRAD_DBG (timing ablations that alter results) are compiled out of the default
    build: the strings are not even in the library; they exist in libkdcc_hip_tuning.so only (make TUNING=1)."""
    from kdcc_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"KDCC_CONV_TUNE", b"KDCC_DW_DBG", b"KDCC_WGRAD_DBG"):
        assert name not in blob, f"{name.decode()} is read by the default library"


def test_kernel_log_api(built):
    from kdcc_amd import _lib
    with _lib.kernel_log() as log:
        pass
    assert log.counts == {} and _lib.last_kernel() == ""


def test_missing_library_is_loud(built, monkeypatch):
    from kdcc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libkdcc_hip.so")
    with pytest.raises(_lib.KdccError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def test_no_compiler_waits_inside_the_conv_main_loops():
    """The persistent conv kernels wait for their LDS-DMA by hand (counted s_waitcnt in inline asm).  hipcc's own wait-count
    insertion must not add an s_waitcnt vmcnt inside those loops -- it does when it believes an epilogue load may still be
    pending at the loop header -- or every stage drains the DMA (measured: 1x1 layers 1.5x slower).  tools/check_loop_waits.py
    compiles conv_igemm.hip to assembly and looks."""
    import shutil
    import subprocess
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_loop_waits.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_hand_scheduled_conv_loop_is_generated_and_audited():
    """csrc/conv_lw_body.inc is the output of tools/gen_conv_lw.py (the hand-scheduled main loop of conv_row_lw_kernel), and
    the compiled kernel passes tools/check_lw_asm.py: hipcc never touches the accumulator file the asm statements own, every
    MFMA sits in the one tile statement, no scratch in the no-operand instantiation."""
    import shutil
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_conv_lw.py"), "--check"])
    assert r.returncode == 0, "csrc/conv_lw_body.inc is stale: run python tools/gen_conv_lw.py"
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lw_asm.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_hand_scheduled_loops_pass_the_schedule_interpreter():
    """tools/check_lw_schedule.py interprets the generated instruction streams of conv_row_lw_kernel / conv_row_tall_kernel for a few
    consecutive tiles (scalar registers, branches, M0, the in-order vmcnt counter, barriers) and checks every fragment read: the
    pieces it finds were fetched from the address the convolution needs there, each of them was covered by a vmcnt wait and a
    barrier before the read, and nothing is staged into a buffer before a barrier behind its last read.  The checker itself is
    held to account with three mutated schedules it has to reject."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_lw_schedule as C
    import gen_conv_lw as G
    lw = lambda: sum((C.check_lw(c, n, stores=s) for c, n in ((64, [3, 2, 3]), (128, [3, 3, 2, 2, 3]), (320, [3, 2, 3, 3])) for s in (32, 64, 96)), [])
    tall = lambda: sum((C.check_tall(c, n, stores=s) for c, n in ((64, [3, 2, 3, 3]), (128, [3, 2, 2, 3]), (192, [3, 3, 2, 3])) for s in (32, 64)), [])
    assert lw() == [] and tall() == []
    issued, waits, shift = G.ISSUED, G.TALL_WAIT, G.tall_shift
    try:
        G.ISSUED = [x + 1 for x in issued]              # every wait of the 3x3 loop two operations too lax
        assert any("in flight" in f for f in lw())
        G.ISSUED = issued
        G.TALL_WAIT = (13, 5, 13)                        # the 512 x 128 loop's row-buffer wait one piece too lax
        assert any("in flight" in f for f in tall())
        G.TALL_WAIT = waits
        G.tall_shift = lambda tag: [l for l in shift(tag) if not l.startswith("s_mov_b64 " + G.D_SBN1)]      # B's staging iterator not advanced
        assert any("holds data from" in f for f in tall())
    finally:
        G.ISSUED, G.TALL_WAIT, G.tall_shift = issued, waits, shift


def test_weight_gradient_lone_wave_loop_is_generated_interpreted_and_audited():
    """csrc/wgrad_lw_body.inc is the output of tools/gen_wgrad_lw.py (the stage loop of conv_wgrad_lw_kernel, pw_wgrad.hip).
    tools/check_wgrad_lw.py walks the generated stream with the two in-order queues and their counted waits: fragments landed before
    their MFMA, every LDS-DMA piece of a stage landed before the barrier behind which the stage is read, ring slots overwritten only
    behind the barrier that retires them; and audits the compiled kernel (the accumulation file untouched by the compiler, 96 MFMAs in
    one statement, no scratch).  The stream is walked with the zero fill of the row-buffer pieces issued (boundary stages) and skipped
    (interior stages).  Mutations it has to reject: the vector-memory waits one piece too lax; the LDS waits one read too lax (caught
    with the zero fill skipped; with it issued ONE is absorbed by design: the generator does not count a write that may be skipped)."""
    import shutil
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_wgrad_lw.py"), "--check"])
    assert r.returncode == 0, "csrc/wgrad_lw_body.inc is stale: run python tools/gen_wgrad_lw.py"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_wgrad_lw as C
    import gen_wgrad_lw as G
    assert C.interpret(G.build()) == []
    try:
        G.SLACK_VM = 1
        assert any("in flight at the barrier" in f for f in C.interpret(G.build()))
        G.SLACK_VM, G.SLACK_DS = 0, 1
        assert any("ds_read_b64_tr_b16" in f and "in flight" in f for f in C.interpret(G.build()))
    finally:
        G.SLACK_DS = G.SLACK_VM = 0
    # the 1x1 form (conv_wgrad_pw_lw_kernel, tools/gen_wgrad_pw_lw.py): same checks, same mutations
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_wgrad_pw_lw.py"), "--check"])
    assert r.returncode == 0, "csrc/wgrad_pw_lw_body.inc is stale: run python tools/gen_wgrad_pw_lw.py"
    import gen_wgrad_pw_lw as GP
    assert C.interpret_pw(GP.build()) == []
    try:
        GP.SLACK_VM = 1
        assert any("in flight at the barrier" in f for f in C.interpret_pw(GP.build()))
        GP.SLACK_VM, GP.SLACK_DS = 0, 1
        assert any("ds_read_b64_tr_b16" in f and "in flight" in f for f in C.interpret_pw(GP.build()))
    finally:
        GP.SLACK_DS = GP.SLACK_VM = 0
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_wgrad_lw.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_depthwise_lone_wave_loop_is_generated_interpreted_and_audited():
    """csrc/dw_lw_body.inc is the output of tools/gen_dw_lw.py (the item loop of dw_lw_fan3_kernel, dwconv_lw.hip).  tools/check_dw_lw.py
    interprets the generated stream with the hardware's two in-order queues (LDS, vector memory) and its counted waits -- every MFMA
    fragment, staged output and transposed tile register must have landed before it is read, every barrier must find the LDS queue
    empty -- and audits the compiled kernel (no compiler instruction in the accumulation file once the operands sit there, all 336
    MFMAs in the one statement, no scratch, v167 the only vector input).  The interpreter is held to account: waits made one
    operation too lax, in either queue, must be reported."""
    import shutil
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_dw_lw.py"), "--check"])
    assert r.returncode == 0, "csrc/dw_lw_body.inc is stale: run python tools/gen_dw_lw.py"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dw_lw as C
    import gen_dw_lw as G
    assert C.interpret(G.build()) == []
    try:
        G.SLACK_DS = 1
        assert any("in flight" in f for f in C.interpret(G.build()))
        G.SLACK_DS, G.SLACK_VM = 0, 1
        assert any("buffer_load" in f and "in flight" in f for f in C.interpret(G.build()))
    finally:
        G.SLACK_DS = G.SLACK_VM = 0
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dw_lw.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
