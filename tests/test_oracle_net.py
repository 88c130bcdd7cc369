"""Pins the network-level oracle (oracle/net_ref.py) against the reference's own student-step outputs. CPU only."""
import numpy as np
import torch

from _netutil import seeded_cheap_weights, seeded_teacher_sd
from _seeded import sample_idx, seeded_input
from oracle import net_ref


def _cmp(t, g, key, tol=2e-5):
    f = t.detach().float().contiguous().reshape(-1)
    assert list(t.shape) == [int(v) for v in g[f"{key}.shape"]]
    ref = g[f"{key}.sample"].astype(np.float64)
    got = f[sample_idx(f.numel())].numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= tol * max(np.abs(ref).max(), 1e-12), key


def test_net_oracle_matches_reference_step(golden):
    g = golden("student_step_g4")
    plan = [str(s) for s in g["plan"]]
    torch.set_num_threads(8)
    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, plan, seeded_cheap_weights(tsd, plan))
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128))
    tgt = torch.from_numpy(g["target"].astype(np.int64))
    r = net_ref.kd_step(tsd, ssd, x, tgt, plan)
    np.testing.assert_allclose(r["hint_loss"].item(), float(g["hint_loss"]), rtol=1e-5)
    np.testing.assert_allclose([p.item() for p in r["per_hint"]], g["per_hint"], rtol=1e-5)
    np.testing.assert_allclose(r["kd_loss"].item(), float(g["kd_loss"]), rtol=1e-4)
    np.testing.assert_allclose(r["kd_mse"].item(), float(g["kd_mse"]), rtol=1e-4)
    np.testing.assert_allclose(r["supervised_loss"].item(), float(g["supervised_loss"]), rtol=1e-5)
    np.testing.assert_allclose(r["teacher_loss"].item(), float(g["teacher_loss"]), rtol=1e-5)
    _cmp(r["student_logits"], g, "student_logits")
    _cmp(r["teacher_logits"], g, "teacher_logits")
    assert r["hint_names"] == plan  # forward-execution order == plan order for this plan
    for i in range(len(plan)):
        _cmp(r["student_hints"][i], g, f"hint_s{i}")
        _cmp(r["teacher_hints"][i], g, f"hint_t{i}")
    assert sorted(r["grads"]) == sorted(str(s) for s in g["trainable"])
    for n, gr in r["grads"].items():
        _cmp(gr, g, f"grad:{n}", tol=1e-4)


def _cmp_k(t, g, key, k, tol):
    f = t.detach().float().contiguous().reshape(-1)
    assert list(t.shape) == [int(v) for v in g[f"{key}.shape"]], key
    ref = g[f"{key}.sample"].astype(np.float64)
    got = f[sample_idx(f.numel(), k)].numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= tol * max(np.abs(ref).max(), 1e-12), key
    ssq = float((f.double() ** 2).sum())
    assert abs(ssq - float(g[f"{key}.sumsq"][0])) <= 4 * tol * float(g[f"{key}.sumsq"][0]) + 1e-30, key


def test_net_oracle_mode_b_matches_reference(golden):
    """loss = KLDiv + hints with EVERY student parameter trainable (dense convs, eval-mode BN weights / biases, the stem),
    hints on `convs` / a whole block / the `aspp` module: net_ref reproduces the reference's losses and all gradients."""
    g = golden("student_step_full_g4")
    plan, hints = [str(s) for s in g["plan"]], [str(s) for s in g["hints"]]
    torch.set_num_threads(8)
    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, plan, seeded_cheap_weights(tsd, plan), trainable="all")
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128))
    r = net_ref.kd_step(tsd, ssd, x, None, plan, backprop="kd+hint", hint_names=hints)
    assert r["hint_names"] == hints
    np.testing.assert_allclose(r["hint_loss"].item(), float(g["hint_loss"]), rtol=1e-5)
    np.testing.assert_allclose([p.item() for p in r["per_hint"]], g["per_hint"], rtol=1e-5)
    np.testing.assert_allclose(r["kd_loss"].item(), float(g["kd_loss"]), rtol=1e-4)
    _cmp_k(r["student_logits"], g, "student_logits", 1024, 2e-5)
    for i in range(len(hints)):
        _cmp_k(r["student_hints"][i], g, f"hint_s{i}", 512, 2e-5)
    names = [str(s) for s in g["trainable"]]
    assert sorted(r["grads"]) == sorted(names) and len(names) == 143
    for n in names:
        _cmp_k(r["grads"][n], g, f"grad:{n}", 256, 2e-4)


def test_taylor_importance_matches_reference(golden):
    """net_ref.taylor_importance (unit gates behind three convs, supervised loss) vs the reference's TaylorPruneStudent run."""
    g = golden("taylor")
    names = [str(s) for s in g["names"]]
    torch.set_num_threads(8)
    tsd = seeded_teacher_sd()
    x = seeded_input("taylor.x", (2, 3, 64, 128))
    tgt = torch.from_numpy(g["target"].astype(np.int64))
    loss, gg, imp = net_ref.taylor_importance(tsd, x, tgt, {n: len(g[f"imp:{n}"]) for n in names})
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    for n in names:
        ref = g[f"gate_grad:{n}"].astype(np.float64)
        assert np.abs(gg[n].numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), n
        np.testing.assert_allclose(imp[n].numpy(), g[f"imp:{n}"], rtol=2e-3, atol=1e-6 * g[f"imp:{n}"].max())


def test_taylor_gate_sites_and_values_match_reference(golden):
    """Gates away from 1 at every site class cfg/taylor_importance_track.json uses (block conv, bnrelu ReLUs, ASPP branch ReLU
    and conv): the oracle's loss and gate gradients vs the reference's TaylorPruneStudent (tests/golden/taylor_steps.npz, `rnd`),
    and -- with unit gates -- the first batch of the reference's three-step runs."""
    g = golden("taylor_steps")
    names = [str(s) for s in g["names"]]
    torch.set_num_threads(8)
    tsd = seeded_teacher_sd()
    x = seeded_input("taylor.steps.x0", (2, 3, 64, 128))
    tgt = torch.from_numpy(g["target0"].astype(np.int64))
    sizes = {n: len(g[f"rnd.gate:{n}"]) for n in names}
    vals = {n: torch.from_numpy(g[f"rnd.gate:{n}"]) for n in names}
    loss, gg, imp = net_ref.taylor_importance(tsd, x, tgt, sizes, values=vals)
    np.testing.assert_allclose(loss.item(), float(g["rnd.loss"]), rtol=1e-5)
    for n in names:
        ref = g[f"rnd.grad:{n}"].astype(np.float64)
        assert np.abs(gg[n].numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), n
        np.testing.assert_allclose(imp[n].numpy(), g[f"rnd.imp:{n}"], rtol=2e-3, atol=1e-6 * g[f"rnd.imp:{n}"].max())
    loss, gg, _ = net_ref.taylor_importance(tsd, x, tgt, sizes)
    np.testing.assert_allclose(loss.item(), float(g["acc1.loss0"]), rtol=1e-5)
    for n in names:
        ref = g[f"acc1.grad0:{n}"].astype(np.float64)
        assert np.abs(gg[n].numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), n


def test_gscnn_oracle_matches_reference(golden):
    """oracle/net_ref.gscnn_forward (trunk + shape stream + gated convs + edge-aware ASPP + decoder, Canny map given) vs the
    reference's GSCNN(19).forward on the same seeded weights / inputs; plus the two building blocks alone."""
    from _netutil import canny_stub_map, seeded_gscnn_sd
    from _seeded import seeded_value
    g = golden("gscnn")
    torch.set_num_threads(8)
    sd = seeded_gscnn_sd()
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128), scale=float(g["x_scale"]))
    canny = torch.stack([canny_stub_map((64, 128), int(s)) for s in g["canny_seeds"]]).unsqueeze(1)
    with torch.no_grad():
        logits, _, _, aux = net_ref.gscnn_forward(sd, x, canny, want_aux=True)
    _cmp(logits, g, "logits")
    _cmp(aux["acts"], g, "acts")
    _cmp(aux["aspp"], g, "aspp")
    _cmp(aux["gate1"], g, "gate1")
    # building blocks
    def blk_sd(prefix, keys):
        return {k: seeded_value(prefix + k, torch.empty(shape)) for k, shape in keys.items()}
    gk = {"weight": (16, 16, 1, 1), "_gate_conv.0.weight": (17,), "_gate_conv.0.bias": (17,), "_gate_conv.0.running_mean": (17,),
          "_gate_conv.0.running_var": (17,), "_gate_conv.1.weight": (17, 17, 1, 1), "_gate_conv.1.bias": (17,),
          "_gate_conv.3.weight": (1, 17, 1, 1), "_gate_conv.3.bias": (1,), "_gate_conv.4.weight": (1,), "_gate_conv.4.bias": (1,),
          "_gate_conv.4.running_mean": (1,), "_gate_conv.4.running_var": (1,)}
    gsd = {"g." + k: v for k, v in blk_sd("gscnn.blk.gate.", gk).items()}
    y = net_ref._gated_conv(gsd, "g", seeded_input("gscnn.blk.gate.f", (2, 16, 12, 20)), seeded_input("gscnn.blk.gate.a", (2, 1, 12, 20)))
    np.testing.assert_allclose(y.numpy(), g["blk_gate.y"], rtol=1e-4, atol=1e-5)
    rk = {"conv1.weight": (16, 16, 3, 3), "conv2.weight": (16, 16, 3, 3)}
    for b in ("bn1", "bn2"):
        for s_ in ("weight", "bias", "running_mean", "running_var"):
            rk[f"{b}.{s_}"] = (16,)
    rsd = {"r." + k: v for k, v in blk_sd("gscnn.blk.res.", rk).items()}
    y = net_ref._basic_block(rsd, "r", seeded_input("gscnn.blk.res.x", (2, 16, 12, 20)))
    np.testing.assert_allclose(y.numpy(), g["blk_res.y"], rtol=1e-4, atol=1e-5)
