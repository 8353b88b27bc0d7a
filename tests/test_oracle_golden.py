"""CPU: the oracle (oracle/dhaug_oracle.py) against the golden vectors captured from the reference.
This is what pins the oracle; the GPU parity tests then compare the HIP path with the oracle."""
import numpy as np
import pytest
import torch

import golden_util as GU
from oracle import dhaug_oracle as O


def maxabs(a, b):
    return (a.double() - b.double()).abs().max().item()


@pytest.mark.parametrize("name", ["fk_N1", "fk_N8", "fk_N1024", "fk_single_dof", "fk_video_B16_R9"])
def test_fk_torch_branch_bit_exact(golden, name):
    g = golden(name)
    out = O.fk_forward32(g["angles"], g["bone_len"], g["root"].reshape(-1, 3))
    # same ATen ops in the same order as R/models_Fk_GAN/forward_kinematics_DH_model.py:562-822
    assert maxabs(out, g["out32"]) == 0.0
    # rows of the 32 that the reference never writes stay at root
    unused = [i for i in range(32) if i not in (0, 1, 2, 3, 6, 7, 8, 12, 13, 14, 15, 17, 18, 19, 25, 26, 27)]
    assert maxabs(out[:, unused], g["root"].reshape(-1, 1, 3).expand(-1, len(unused), -1)) == 0.0


def test_fk_fp64_agrees(golden):
    g = golden("fk_N1024")
    out64 = O.fk_forward32(g["angles"].double(), g["bone_len"].double(), g["root"].double())
    assert maxabs(out64, g["out32"]) < 2e-6      # fp32 rounding of the reference itself


def test_fk_numpy_branch_and_tpose(golden):
    g = golden("fk_numpy_branch")
    t = O.fk_scalar_numpy(np.zeros(37), O.TPOSE_BONE_LEN, (0.0, 0.0, 0.0))
    assert np.abs(t - g["tpose32"].numpy()).max() < 1e-7
    # known answers, SURVEY.md section 4
    for j, xyz in ((1, (0.25, 0, 0)), (3, (0.25, 0, -1.1)), (13, (0, 0, 0.45)), (15, (0, 0, 0.6)), (19, (-0.4, 0, -0.3))):
        assert np.allclose(t[j], xyz, atol=1e-6), (j, t[j])
    for i in range(4):
        o = O.fk_scalar_numpy(g["angles"][i].numpy(), g["bone_len"][i].numpy(), g["root"][i].numpy())
        assert np.abs(o - g["out32"][i].numpy()).max() < 1e-6


def test_fk_properties(golden):
    g = golden("fk_N1024")
    a, bl, rt = g["angles"], g["bone_len"], g["root"]
    out = O.fk_forward16(a, bl, rt)
    # bone-length invariance
    assert maxabs(O.bone_lengths(out), bl) < 2e-6
    # leaf-joint angles and slot 33 have no effect
    a2 = a.clone()
    a2[:, [4, 9, 22, 27, 32, 33]] += 77.0
    assert maxabs(O.fk_forward16(a2, bl, rt), out) < 1e-6


def test_dh_and_rotation(golden):
    g = golden("dh_rot_16")
    assert maxabs(O.dh_matrix(g["alpha"], g["a"], g["d"], g["theta"]), g["dh"]) == 0.0
    assert maxabs(O.rotation_matrix(g["ang3"][:, 0], g["ang3"][:, 1], g["ang3"][:, 2]), g["rot"]) == 0.0


def test_kcs(golden):
    g = golden("kcs_256")
    assert maxabs(O.bone_vectors(g["pose16"]), g["bonevec"]) < 1e-6
    assert maxabs(O.kcs_features(g["pose16"]), g["kcs30"]) < 2e-6
    assert maxabs(O.kcs_features(g["pose16"], with_lengths=False), g["kcs15"]) < 2e-6


@pytest.mark.parametrize("D,suffix", [(32, ""), (256, ""), (256, "_s2")])
def test_generator(golden, D, suffix):
    g = golden("gen_D%d%s" % (D, suffix))
    sd = GU.seeded_state_dict(GU.shapes_generator(D), int(g["weight_seed"]))
    assert maxabs(O.bone_lengths(g["real16"]), g["bone_len"]) < 1e-6
    head = O.gen_trunk(g["z"], sd)
    assert maxabs(head, g["head"]) < 1e-5
    # tail from the reference's own head output: isolates tanh/scatter/limits/jitter/FK
    fake, ang = O.gen_tail(g["head"], g["bone_len"], g["scaler"])
    assert maxabs(ang, g["angle37"]) == 0.0
    assert maxabs(fake, g["fake"]) == 0.0
    fake2, _, _ = O.generator_forward(g["z"], sd, g["bone_len"], g["scaler"])
    assert maxabs(fake2, g["fake"]) < 1e-4


def test_generator_no_preangle(golden):
    g, n = golden("gen_D32"), golden("gen_D32_nopre")
    fake, ang = O.gen_tail(g["head"], g["bone_len"], g["scaler"], use_preangle=False)
    assert maxabs(ang, n["angle37"]) == 0.0 and maxabs(fake, n["fake"]) == 0.0


def test_video_generator(golden):
    g = golden("gen_video_D32")
    sd = GU.seeded_state_dict(GU.shapes_generator(32, frames=9), int(g["weight_seed"]))
    assert maxabs(O.bone_lengths(g["real16"]), g["bone_len"]) < 1e-6
    fake, _, ang = O.generator_forward(g["z"], sd, g["bone_len"], g["scaler"], frames=9)
    assert fake.shape == (8, 9, 48)
    assert maxabs(ang, g["angle37"]) < 2e-4          # degrees, through the fp32 trunk
    assert maxabs(fake, g["fake"]) < 1e-5


@pytest.mark.parametrize("D,suffix", [(32, ""), (256, ""), (256, "_s2")])
def test_critics(golden, D, suffix):
    g = golden("critics_D%d%s" % (D, suffix))
    sd3 = GU.seeded_state_dict(GU.shapes_d3(D), int(g["weight_seed3"]))
    sd2 = GU.seeded_state_dict(GU.shapes_d2(D), int(g["weight_seed2"]))
    l3, l2 = O.d3_forward(g["x3"], sd3), O.d2_forward(g["x2"], sd2)
    assert ((l3 - g["logit3"]).abs() / g["logit3"].abs().clamp_min(1e-3)).max() < 1e-4
    assert ((l2 - g["logit2"]).abs() / g["logit2"].abs().clamp_min(1e-3)).max() < 1e-4


def _motion_shapes(ref_keys_fn):
    return ref_keys_fn


def test_motion_critics(golden):
    g = golden("motion_critics_D32")
    D, R = 32, 9
    s3 = {}
    for name, width in (("special_KCS", R * 15), ("diff_special_KCS", (R - 1) * 15), ("pos_3d", R * 48),
                        ("diff_pos_3d", (R - 1) * 48)):
        s3[name + "_previous.0.weight"] = (D, width); s3[name + "_previous.0.bias"] = (D,)
        for i in (1, 2, 3):
            GU._res(s3, "%s_block%d" % (name, i), D)
    s3["kcs_merge_previous.0.weight"] = (100, 4 * D); s3["kcs_merge_previous.0.bias"] = (100,)
    GU._res(s3, "kcs_merge_block1", 100)
    s3["kcs_output.weight"] = (1, 100); s3["kcs_output.bias"] = (1,)
    s2 = {}
    for name, width in (("pos_2d", R * 32), ("root_diff_2d", (R - 1) * 2)):
        s2[name + "_previous.0.weight"] = (D, width); s2[name + "_previous.0.bias"] = (D,)
        for i in (1, 2, 3):
            GU._res(s2, "%s_block%d" % (name, i), D)
    s2["merge_previous.0.weight"] = (100, 2 * D); s2["merge_previous.0.bias"] = (100,)
    GU._res(s2, "merge_block1", 100)
    s2["merge_output.weight"] = (1, 100); s2["merge_output.bias"] = (1,)
    sd3 = GU.seeded_state_dict(s3, int(g["weight_seed3"]))
    sd2 = GU.seeded_state_dict(s2, int(g["weight_seed2"]))
    l3 = O.motion_d3_forward(g["x3"], sd3, R)
    l2 = O.motion_d2_forward(g["x2"], sd2, R)
    assert l3.shape == (8, 1) and l2.shape == (8, 1)
    assert ((l3 - g["logit3"]).abs() / g["logit3"].abs().clamp_min(1e-3)).max() < 1e-4
    assert ((l2 - g["logit2"]).abs() / g["logit2"].abs().clamp_min(1e-3)).max() < 1e-4


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_gradient_penalty_and_critic_step(golden, tag):
    g = golden("critic_step_%s_D32" % tag)
    shapes = GU.shapes_d3(32) if tag == "d3" else GU.shapes_d2(32)
    fwd = O.d3_forward if tag == "d3" else O.d2_forward
    sd = GU.seeded_state_dict(shapes, int(g["weight_seed"]))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    gp = O.gradient_penalty(lambda x: fwd(x, params), g["real"], g["fake"], g["alpha"], 10.0)
    assert abs(gp.item() - g["gp"].item()) <= 1e-5 * max(1.0, abs(g["gp"].item()))
    grads = torch.autograd.grad(gp, list(params.values()), allow_unused=True)
    for (k, _), gr in zip(params.items(), grads):
        ref = g["gpgrad__" + k]
        gr = torch.zeros_like(ref) if gr is None else gr
        assert maxabs(gr, ref) <= 1e-5 + 1e-4 * ref.abs().max().item(), k
    r = O.critic_step(fwd, sd, g["real"], g["fake"], g["alpha"])
    assert abs(r["Wasserstein_D"].item() - g["Wasserstein_D"].item()) < 1e-6
    assert abs(r["D_cost"].item() - g["D_cost"].item()) <= 1e-5 * max(1.0, abs(g["D_cost"].item()))
    for k in sd:
        assert maxabs(r["grads"][k], g["grad__" + k]) <= 1e-5 + 1e-4 * g["grad__" + k].abs().max().item(), k
        assert maxabs(r["new_params"][k], g["new__" + k]) <= 2e-6, k      # one Adam step, lr 1e-4


def _ref_record(g, kind, k):
    return {part: g["%s__%s__%s" % (kind, part, k)] for part in ("full", "sample", "proj") if "%s__%s__%s" % (kind, part, k) in g}


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_critic_step_at_dense_dim_256(golden, tag):
    """the oracle's critic step against the reference's train_Fk_discriminator at the benchmark's width (compact records of
    the 0.9 M / 0.27 M gradients: tests/golden/make_golden_d256.py)"""
    g = golden("critic_step_%s_D256" % tag)
    shapes = GU.shapes_d3(256) if tag == "d3" else GU.shapes_d2(256)
    fwd = O.d3_forward if tag == "d3" else O.d2_forward
    sd = GU.seeded_state_dict(shapes, int(g["weight_seed"]))
    r = O.critic_step(fwd, sd, g["real"], g["fake"], g["alpha"])
    assert abs(r["Wasserstein_D"].item() - g["Wasserstein_D"].item()) < 2e-6
    assert abs(r["D_cost"].item() - g["D_cost"].item()) <= 1e-5 * max(1.0, abs(g["D_cost"].item()))
    for i, k in enumerate(sd):
        GU.compact_close(r["grads"][k], _ref_record(g, "grad", k), 100 + i, 1e-6, 2e-4, k)


def test_camera(golden):
    g = golden("camera_128")
    Xc = O.world_to_camera(g["X"], g["R"], g["t"])
    assert maxabs(Xc, g["Xc"]) < 1e-6
    assert maxabs(O.project_to_2d(g["Xc"], g["cam"]), g["x2d"]) < 1e-6
    assert maxabs(O.camera_to_world(g["Xc"], g["R"].repeat(128, 1), g["t"].repeat(128, 1)), g["Xw"]) < 1e-6
    assert maxabs(O.flip_lr(g["X"]), g["flip"]) == 0.0


def test_random_bl_aug(golden):
    import json, os
    g = golden("bl_aug_96")
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd")
    T = torch.tensor(json.load(open(os.path.join(pkg, "common", "bl_templates.json")))["templates"])
    assert T.shape == (5, 15)
    out = O.random_bl_aug(g["x"], T[g["idx"].long()])
    assert maxabs(out, g["out"]) < 2e-6
    assert maxabs(O.project_to_2d(g["out"], g["cam"]), g["proj"]) < 1e-6


def test_fk_op_by_op_variant_is_bit_identical(golden):
    """the reference-granularity FK restatement (bench.py's cpu_baseline_faithful) gives exactly the batched restatement's
    values -- and therefore the reference's (fk_N1024 golden)"""
    g = golden("fk_N1024")
    out = O.fk_forward32_op_by_op(g["angles"], g["bone_len"], g["root"].reshape(-1, 3))
    assert maxabs(out, g["out32"]) == 0.0
