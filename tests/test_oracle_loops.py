"""CPU: the oracle's restatement of the epoch loops (oracle.gan_iteration / video_gan_iteration / critic_step_net) against
golden vectors captured from the REFERENCE's own GAN_solutions_FK_generator / video_mode_GAN_solutions_FK_generator runs
(tests/golden/make_golden_loops.py, recorded random draws replayed).  Pins rows a16 / a18 of SURVEY.md section 8."""
import pytest
import torch

import golden_util as GU
import loop_util as LU
from oracle import dhaug_oracle as O


def maxabs(a, b):
    return (a.double() - b.double()).abs().max().item()


def test_single_frame_loop_golden(golden):
    g = golden("gan_loop_D32")
    r = LU.replay_single_oracle(g)
    # the fake pairs of all five iterations (generator weights are the seeded ones until the G step of the 5th)
    assert maxabs(r["buf_p3"], g["buf_p3"]) <= 2e-6 and maxabs(r["buf_p2"], g["buf_p2"]) <= 2e-6
    # critic scalars of every step, in the reference's logging order
    for name, ref in LU.scalar_series(g).items():
        got = torch.tensor(r["scalars"][name], dtype=torch.float64)
        assert got.shape == ref.shape, name
        assert maxabs(got, ref) <= 2e-5 * max(1.0, ref.abs().max().item()), name
    # the G step: critic weights at that moment, the generator's gradients, its weights after Adam
    for k in LU.keys(g, "gstep_d3__"):
        assert maxabs(r["gstep_d"]["d3"][k], g["gstep_d3__" + k]) <= 2e-6, k
    for k in LU.keys(g, "gstep_grad__"):
        ref = g["gstep_grad__" + k]
        assert maxabs(r["g_grads"][k], ref) <= 1e-7 + 2e-4 * ref.abs().max().item(), k
    for tag, net in (("final_G__", r["G"]), ("final_d3__", r["D3"]), ("final_d2__", r["D2"])):
        for k in LU.keys(g, tag):
            assert maxabs(net.state()[k], g[tag + k]) <= 5e-6, (tag, k)


@pytest.mark.parametrize("D", [256, 1000])
def test_single_frame_loop_golden_at_dense_dim_256(golden, D):
    """the same loop at the width of the reference's README command and of the benchmark (gan_loop_D256: weights and gradients
    as compact records -- strided samples + seeded +-1 projections -- of their change from the seeded initial weights), and at the
    reference's DEFAULT width (gan_loop_D1000, B = 128: R/function_aug/config.py:101-109)"""
    g = golden("gan_loop_D%d" % D)
    r = LU.replay_single_oracle(g, D)
    assert maxabs(r["buf_p3"], g["buf_p3"]) <= 5e-6 and maxabs(r["buf_p2"], g["buf_p2"]) <= 5e-6
    for name, ref in LU.scalar_series(g).items():
        got = torch.tensor(r["scalars"][name], dtype=torch.float64)
        assert got.shape == ref.shape, name
        assert maxabs(got, ref) <= 5e-5 * max(1.0, ref.abs().max().item()), name
    init = dict(zip(("G", "d3", "d2"), LU.single_state_dicts(g, D)))
    names = {t: list(init[t]) for t in init}
    for i, k in enumerate(names["G"]):
        ref = LU.compact_record(g, "gstep_grad__", k)
        # (D = 1000: measured 1.2e-3 of the tensor's scale at worst -- the critics the G step differentiates through have taken ten
        # sign-like Adam steps, and with 12 M parameters more elements sit within rounding of a zero gradient; pairs, scalars and
        # the critics' weights agree as at D = 256)
        GU.compact_close(r["g_grads"][k], ref, 100 + i, 1e-7, 5e-4 if D == 256 else 2.5e-3, "gstep_grad " + k)
    # weights: the CHANGE over the loop (10 Adam steps of 1e-4 for the critics, 1 for the generator); an element whose gradient
    # is within rounding of zero may step the other way (lr * g / (|g| + eps)), hence the hard bound of (steps x lr)
    for tag, net, prefix, steps in (("G", r["G"], "final_G__", 1), ("d3", r["D3"], "final_d3__", 10), ("d2", r["D2"], "final_d2__", 10)):
        for i, k in enumerate(names[tag]):
            ref = LU.compact_record(g, prefix, k)
            got = GU.compact(net.state()[k] - init[tag][k], 100 + i)
            key = "full" if "full" in ref else "sample"
            e = (got[key].double() - ref[key].double()).abs()
            assert e.max().item() <= 2.05 * steps * 1e-4, (tag, k, e.max().item())
            assert torch.quantile(e, 0.98).item() <= 2e-5, (tag, k, torch.quantile(e, 0.98).item())


def test_video_loop_golden(golden):
    g = golden("video_loop_D32")
    r = LU.replay_video_oracle(g)
    assert maxabs(r["buf_p3"], g["buf_p3"]) <= 2e-6 and maxabs(r["buf_p2"], g["buf_p2"]) <= 2e-6
    for name, ref in LU.scalar_series(g).items():
        got = torch.tensor(r["scalars"][name], dtype=torch.float64)
        assert got.shape == ref.shape, name
        assert maxabs(got, ref) <= 5e-5 * max(1.0, ref.abs().max().item()), name
    for k in LU.keys(g, "gstep_grad__"):
        ref = g["gstep_grad__" + k]
        assert maxabs(r["g_grads"][k], ref) <= 1e-7 + 5e-4 * ref.abs().max().item(), k
    for tag in ("G", "d3", "d2", "m3", "m2"):
        for k in LU.keys(g, "final_%s__" % tag):
            assert maxabs(r["nets"][tag].state()[k], g["final_%s__%s" % (tag, k)]) <= 1e-5, (tag, k)


@pytest.mark.parametrize("tag", ["m3", "m2"])
def test_motion_critic_step_golden(golden, tag):
    """one train_Fk_discriminator call per motion critic in the mode the video loop uses for it: GP over B clips for the
    3D motion critic (dis_mode='motion'), over B*R frames for the 2D one (default mode, R/...video_GAN_fun.py:341-346)"""
    g = golden("motion_step_%s_D32" % tag)
    B, R = 8, 9
    sd = GU.seeded_state_dict(LU.motion_shapes(32, R)[0 if tag == "m3" else 1], int(g["weight_seed"]))
    fwd = (lambda x, p: O.motion_d3_forward(x, p, R)) if tag == "m3" else (lambda x, p: O.motion_d2_forward(x, p, R))
    net = O.Net(sd, fwd)
    rows = B if tag == "m3" else B * R
    assert g["alpha"].shape == (rows, 1)
    W, C = O.critic_step_net(net, g["real"], g["fake"], g["alpha"], rows)
    # grads are read after the step: recompute them on a fresh net
    net2 = O.Net(sd, fwd)
    net2.zero_grad()
    gp = O.gradient_penalty(net2, g["real"].reshape(rows, -1), g["fake"].reshape(rows, -1), g["alpha"])
    (net2(g["fake"]).mean() - net2(g["real"]).mean() + gp).backward()
    assert abs(W.item() - g["Wasserstein_D"].item()) <= 1e-6 and abs(C.item() - g["D_cost"].item()) <= 1e-5
    for k, gr in net2.grads().items():
        ref = g["grad__" + k]
        assert maxabs(gr, ref) <= 1e-7 + 1e-4 * ref.abs().max().item(), k
    for k, v in net.state().items():
        assert maxabs(v, g["new__" + k]) <= 2e-6, k


@pytest.mark.parametrize("tag", ["m3", "m2"])
def test_motion_critic_step_golden_at_dense_dim_1000(golden, tag):
    """the same at the reference's DEFAULT width (DenseDim 1000, README video command; 25.5 M / 12.5 M parameters, B = 128 clips):
    scalars, every gradient (compact records: strided samples + seeded +-1 projections) and the Adam update"""
    g = golden("motion_step_%s_D1000" % tag)
    B, R, D = 128, 9, 1000
    shapes = LU.motion_shapes(D, R)[0 if tag == "m3" else 1]
    sd = GU.seeded_state_dict(shapes, int(g["weight_seed"]))
    fwd = (lambda x, p: O.motion_d3_forward(x, p, R)) if tag == "m3" else (lambda x, p: O.motion_d2_forward(x, p, R))
    rows = B if tag == "m3" else B * R
    assert g["alpha"].shape == (rows, 1)
    net = O.Net(sd, fwd)
    net.zero_grad()
    gp = O.gradient_penalty(net, g["real"].reshape(rows, -1), g["fake"].reshape(rows, -1), g["alpha"])
    (net(g["fake"]).mean() - net(g["real"]).mean() + gp).backward()
    grads = net.grads()
    net2 = O.Net(sd, fwd)
    W, C = O.critic_step_net(net2, g["real"], g["fake"], g["alpha"], rows)
    assert abs(W.item() - g["Wasserstein_D"].item()) <= 2e-6 and abs(C.item() - g["D_cost"].item()) <= 1e-5 * max(1.0, abs(g["D_cost"].item()))
    rec = lambda kind, k: {part: g["%s__%s__%s" % (kind, part, k)] for part in ("full", "sample", "proj") if "%s__%s__%s" % (kind, part, k) in g}
    for i, k in enumerate(sd):
        GU.compact_close(grads[k], rec("grad", k), 100 + i, 1e-7, 2e-4, k)
        ref, got = rec("delta", k), GU.compact(net2.state()[k] - sd[k], 100 + i)
        key = "full" if "full" in ref else "sample"
        gref = rec("grad", k)[key]
        well = gref.abs() > max(1e-3 * gref.abs().max().item(), 1e-7)     # (Adam's first step is lr * sign-like: where |g| is not within rounding of 0)
        if well.any():
            assert (got[key].double() - ref[key].double())[well].abs().max().item() <= 2e-6, k
        assert (got[key].double() - ref[key].double()).abs().max().item() <= 2.01e-4, k


def test_video_D1000_forward_golden(golden):
    """BASELINE configs[4] widths (DenseDim 1000, R = 9): the oracle's video generator and four critics against the reference"""
    g = golden("video_D1000")
    R, D = 9, 1000
    sG, s3, s2, sm3, sm2 = (int(v) for v in g["seeds"])
    m3s, m2s = LU.motion_shapes(D, R)
    sd = dict(G=GU.seeded_state_dict(GU.shapes_generator(D, frames=R), sG), d3=GU.seeded_state_dict(GU.shapes_d3(D), s3),
              d2=GU.seeded_state_dict(GU.shapes_d2(D), s2), m3=GU.seeded_state_dict(m3s, sm3), m2=GU.seeded_state_dict(m2s, sm2))
    fake, _, _ = O.generator_forward(g["z"], sd["G"], g["bone_len"], g["scaler"], frames=R)
    # (K = 1000 dot products: the summation order of the CPU GEMM depends on the thread count -- 1e-6 of the +-10 m root range)
    assert maxabs(fake, g["fake"]) <= 1e-5
    rel = lambda a, b: ((a - b).abs() / b.abs().clamp_min(0.1 * b.abs().mean())).max().item()
    assert rel(O.d3_forward(g["x3"], sd["d3"]), g["logit_d3"]) <= 1e-5
    assert rel(O.d2_forward(g["x2"], sd["d2"]), g["logit_d2"]) <= 1e-5
    assert rel(O.motion_d3_forward(g["x3"], sd["m3"], R), g["logit_m3"]) <= 1e-5
    assert rel(O.motion_d2_forward(g["x2"], sd["m2"], R), g["logit_m2"]) <= 1e-5
