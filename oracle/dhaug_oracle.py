"""CPU ORACLE for the DH-AUG FK + GAN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (PyTorch-CPU fp32 / fp64 ATen ops + numpy) of the
reference algorithm for the hot path named in BASELINE.json.  It is the *checker*:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product
path (the dhaug_amd package -> libdhaug.so HIP kernels) never imports, calls or falls back to it.

Parity status: PINNED.  Every function here is checked against golden vectors captured from the
reference's own Python run in the build container (tests/golden/make_golden.py imports the reference
from /root/reference and writes tests/golden/*.npz; tests/test_oracle_golden.py compares).

Citations use R/ = /root/reference/DH-AUG_master/.

Arithmetic notes that matter for parity (all verified against the reference):
  * degrees -> radians exactly as the reference does it in fp32: fp32(fp32(x / 180) * fp32(pi))
    (R/models_Fk_GAN/forward_kinematics_DH_model.py:89-90), so cos(+-90 deg) is -4.371139e-08, not 0.
  * modified-DH (Craig) matrix layout, chain products left-to-right with torch.bmm, translation
    column extraction, global rotation Rx*Ry*Rz applied AFTER the chain, root added last.
  * the arm chains re-use the body chain's first nine local matrices; multiplying them again gives
    bit-identical results to sharing the body cumulative product at index 8.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# Constant tables
# ----------------------------------------------------------------------------------------------
# R/models_Fk_GAN/forward_kinematics_DH_model.py:234-261  (alpha, a, d, theta0) per chain, degrees/metres.
# Length-bearing entries are overwritten per pose (R/...:571-589); the defaults are the T-pose.
RLEG = dict(alpha=[0.0, -90.0, -90.0, 0.0, 0.0], a=[0.25, 0.0, 0.0, 0.6, 0.5], d=[0.0] * 5,
            theta=[0.0, -90.0, 180.0, 0.0, 0.0])
LLEG = dict(alpha=[0.0, 90.0, 90.0, 0.0, 0.0], a=[-0.25, 0.0, 0.0, 0.6, 0.5], d=[0.0] * 5,
            theta=[180.0, -90.0, 0.0, 0.0, 0.0])
BODY = dict(alpha=[0.0] + [-90.0] * 11 + [90.0], a=[0.0] * 12 + [0.15],
            d=[0.0, 0.0, 0.0, 0.25, 0.0, 0.0, 0.2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
            theta=[90.0] + [-90.0] * 10 + [0.0, 0.0])
RARM = dict(alpha=[-90.0, -90.0, -90.0, 0.0, 0.0], a=[-0.3, 0.0, 0.0, 0.4, 0.35], d=[0.0] * 5,
            theta=[-180.0, -90.0, 180.0, 0.0, 0.0])
LARM = dict(alpha=[-90.0, 90.0, 90.0, 0.0, 0.0], a=[0.3, 0.0, 0.0, 0.4, 0.35], d=[0.0] * 5,
            theta=[0.0, -90.0, 0.0, 0.0, 0.0])

# R/common/h36m_dataset.py:37-38
H36M_32_TO_16 = [0, 1, 2, 3, 6, 7, 8, 12, 13, 15, 17, 18, 19, 25, 26, 27]
# R/models_Fk_GAN/forward_kinematics_DH_model.py:46-49 (16-joint indices, FK bone order)
BONE_PAIRS = [(5, 6), (2, 3), (4, 5), (1, 2), (0, 4), (0, 1), (0, 7), (7, 8), (8, 10), (8, 13),
              (10, 11), (13, 14), (11, 12), (14, 15), (8, 9)]
BONE_NAMES = ["left_small_leg", "right_small_leg", "left_big_leg", "right_big_leg", "left_hip", "right_hip",
              "waist", "thorax", "left_shoulder", "right_shoulder", "left_big_arm", "right_big_arm",
              "left_small_arm", "right_small_arm", "neck"]
# R/models_Fk_GAN/Fk_discriminator.py:81-140: KCS cosine pairs (indices into the 15 bones)
KCS_PAIRS = [(0, 2), (1, 3), (2, 4), (3, 5), (4, 5), (4, 6), (5, 6), (6, 7), (7, 14), (7, 8), (7, 9),
             (8, 10), (9, 11), (10, 12), (11, 13)]
# R/models_Fk_GAN/Fk_generator.py:41-76 joint limits (slots 0..33), :35-39 global rotation (slots 34..36)
ANGLE_LO = [-110, -110, -110, -180, 0, -65, -65, -110, -180, 0] + [-180] * 12 + [0, 0] + \
           [-155, -155, -100, 0, 0, -65, -65, -100, 0, 0] + [-180, -180, -180]
ANGLE_HI = [65, 65, 180, 0, 0, 110, 110, 180, 0, 0] + [180] * 12 + [0, 0] + \
           [65, 65, 180, 180, 0, 155, 155, 180, 180, 0] + [180, 180, 180]
ZERO_SLOTS = (4, 9, 22, 23, 28, 33)             # R/models_Fk_GAN/Fk_generator.py:136
LIVE_SLOTS = [i for i in range(37) if i not in ZERO_SLOTS]     # 31 slots <- head columns 0..30
# R/models_Fk_GAN/Fk_generator.py:216-230: FK bone index -> jitter column (-1: thorax, never jittered)
JITTER_COL = [0, 0, 1, 1, 2, 2, 3, -1, 4, 4, 5, 5, 6, 6, 7]
# L/R flip: R/models_Fk_GAN/model_fk_gan_train.py:320-331
FLIP_LEFT = [4, 5, 6, 10, 11, 12]
FLIP_RIGHT = [1, 2, 3, 13, 14, 15]

assert len(ANGLE_LO) == 37 and len(ANGLE_HI) == 37 and len(LIVE_SLOTS) == 31


# ----------------------------------------------------------------------------------------------
# a1 / a2: dh_matrix, rotationMatrix
# ----------------------------------------------------------------------------------------------
def _deg2rad(x):
    """fp32(fp32(x/180) * fp32(pi)) -- R/models_Fk_GAN/forward_kinematics_DH_model.py:89-90."""
    return x / 180 * torch.tensor(np.pi, dtype=x.dtype)


def dh_matrix(alpha, a, d, theta):
    """Modified-DH homogeneous transform, batched.  alpha/a/d/theta: (..., ) degrees / metres.
    R/models_Fk_GAN/forward_kinematics_DH_model.py:80-116."""
    al = _deg2rad(alpha)
    th = _deg2rad(theta)
    ct, st, ca, sa = torch.cos(th), torch.sin(th), torch.cos(al), torch.sin(al)
    z = torch.zeros_like(th)
    o = torch.ones_like(th)
    rows = [ct, -st, z, a + z,
            st * ca, ct * ca, -sa, -sa * d,
            st * sa, ct * sa, ca, ca * d,
            z, z, z, o]
    return torch.stack(rows, dim=-1).reshape(th.shape + (4, 4))


def rotation_matrix(ax, ay, az):
    """Rx(ax) * Ry(ay) * Rz(az), degrees.  R/models_Fk_GAN/forward_kinematics_DH_model.py:141-191."""
    # note: here the reference multiplies by the python float np.pi (not a fp32 tensor); for fp32
    # tensors ATen casts the scalar to fp32, so the arithmetic is the same as _deg2rad.
    ax, ay, az = ax / 180 * np.pi, ay / 180 * np.pi, az / 180 * np.pi
    z, o = torch.zeros_like(ax), torch.ones_like(ax)
    cx, sx, cy, sy, cz, sz = torch.cos(ax), torch.sin(ax), torch.cos(ay), torch.sin(ay), torch.cos(az), torch.sin(az)
    R1 = torch.stack([o, z, z, z, cx, -sx, z, sx, cx], -1).reshape(-1, 3, 3)
    R2 = torch.stack([cy, z, sy, z, o, z, -sy, z, cy], -1).reshape(-1, 3, 3)
    R3 = torch.stack([cz, -sz, z, sz, cz, z, z, z, o], -1).reshape(-1, 3, 3)
    return R1.bmm(R2).bmm(R3)


# ----------------------------------------------------------------------------------------------
# a3 / a4 / a6: forward kinematics
# ----------------------------------------------------------------------------------------------
def _chain_tables(bone_len, dtype):
    """Per-pose (alpha, a, d, theta0) tensors for the five chains with the bone lengths written in.
    bone_len: (N, 15) in FK bone order.  R/models_Fk_GAN/forward_kinematics_DH_model.py:571-589."""
    N = bone_len.shape[0]
    L = {n: bone_len[:, i] for i, n in enumerate(BONE_NAMES)}

    def tab(c):
        return {k: torch.tensor(v, dtype=dtype).repeat(N, 1) for k, v in c.items()}

    rl, ll, bd, ra, la = tab(RLEG), tab(LLEG), tab(BODY), tab(RARM), tab(LARM)
    ll["a"][:, 0] = -L["left_hip"]; ll["a"][:, 3] = L["left_big_leg"]; ll["a"][:, 4] = L["left_small_leg"]
    rl["a"][:, 0] = L["right_hip"]; rl["a"][:, 3] = L["right_big_leg"]; rl["a"][:, 4] = L["right_small_leg"]
    bd["a"][:, 12] = L["neck"]; bd["d"][:, 3] = L["waist"]; bd["d"][:, 6] = L["thorax"]
    la["a"][:, 0] = L["left_shoulder"]; la["a"][:, 3] = L["left_big_arm"]; la["a"][:, 4] = L["left_small_arm"]
    ra["a"][:, 0] = -L["right_shoulder"]; ra["a"][:, 3] = L["right_big_arm"]; ra["a"][:, 4] = L["right_small_arm"]
    return rl, ll, bd, ra, la


def _chain_cumprod(local):
    """local: (N, n, 4, 4) -> cumulative left-to-right products, same shape.
    R/models_Fk_GAN/forward_kinematics_DH_model.py:659-677."""
    out = [local[:, 0]]
    for i in range(1, local.shape[1]):
        out.append(torch.bmm(out[-1], local[:, i]))
    return torch.stack(out, dim=1)


def fk_forward32(angles, bone_len, root):
    """Full FK.  angles (N,37) degrees in generator_angle layout
    [0:5] r-leg, [5:10] l-leg, [10:23] body, [23:28] r-arm, [28:33] l-arm, [33] unused, [34:37] global rot
    (R/models_Fk_GAN/Fk_generator.py:179-186); bone_len (N,15) FK bone order; root (N,3).
    Returns (N,32,3).  R/models_Fk_GAN/forward_kinematics_DH_model.py:562-822."""
    dtype = angles.dtype
    N = angles.shape[0]
    rl, ll, bd, ra, la = _chain_tables(bone_len, dtype)

    def local(tab, ang):
        return dh_matrix(tab["alpha"], tab["a"], tab["d"], tab["theta"] + ang)

    rl_m = local(rl, angles[:, 0:5])
    ll_m = local(ll, angles[:, 5:10])
    bd_m = local(bd, angles[:, 10:23])
    ra_m = torch.cat([bd_m[:, 0:9], local(ra, angles[:, 23:28])], dim=1)       # :629-642
    la_m = torch.cat([bd_m[:, 0:9], local(la, angles[:, 28:33])], dim=1)       # :644-656
    Rg = rotation_matrix(angles[:, 34], angles[:, 35], angles[:, 36])           # :566

    def pos(m):                                                                 # :679-743
        c = _chain_cumprod(m)
        p = c[:, :, 0:3, 3].transpose(1, 2).contiguous()                        # (N,3,n)
        return Rg.bmm(p)

    rl_p, ll_p, bd_p, ra_p, la_p = pos(rl_m), pos(ll_m), pos(bd_m), pos(ra_m), pos(la_m)
    out = torch.zeros((N, 32, 3), dtype=dtype)
    # :751-817
    out[:, 0] = bd_p[:, :, 0]
    out[:, 1] = rl_p[:, :, 0]; out[:, 2] = rl_p[:, :, 3]; out[:, 3] = rl_p[:, :, 4]
    out[:, 6] = ll_p[:, :, 0]; out[:, 7] = ll_p[:, :, 3]; out[:, 8] = ll_p[:, :, 4]
    out[:, 12] = bd_p[:, :, 3]; out[:, 13] = bd_p[:, :, 6]
    out[:, 14] = bd_p[:, :, 12]; out[:, 15] = bd_p[:, :, 12]
    out[:, 17] = la_p[:, :, 9]; out[:, 18] = la_p[:, :, 12]; out[:, 19] = la_p[:, :, 13]
    out[:, 25] = ra_p[:, :, 9]; out[:, 26] = ra_p[:, :, 12]; out[:, 27] = ra_p[:, :, 13]
    return out + root.reshape(-1, 1, 3)                                         # :819-820


def _dh_matrix_slicewise(alpha, a, d, theta):
    """One joint's modified-DH matrix the way the reference builds it: a float64 numpy zeros block turned into an fp32
    tensor, then sixteen strided slice assignments with the sines / cosines re-evaluated per entry
    (R/models_Fk_GAN/forward_kinematics_DH_model.py:89-116).  Same values as dh_matrix()."""
    al = _deg2rad(alpha)
    th = _deg2rad(theta)
    m = torch.tensor(np.zeros((theta.shape[0], 4, 4)), dtype=torch.float32)
    m[:, 0, 0] = torch.cos(th); m[:, 0, 1] = -torch.sin(th); m[:, 0, 2] = 0; m[:, 0, 3] = a
    m[:, 1, 0] = torch.sin(th) * torch.cos(al); m[:, 1, 1] = torch.cos(th) * torch.cos(al)
    m[:, 1, 2] = -torch.sin(al); m[:, 1, 3] = -torch.sin(al) * d
    m[:, 2, 0] = torch.sin(th) * torch.sin(al); m[:, 2, 1] = torch.cos(th) * torch.sin(al)
    m[:, 2, 2] = torch.cos(al); m[:, 2, 3] = torch.cos(al) * d
    m[:, 3, 0] = 0; m[:, 3, 1] = 0; m[:, 3, 2] = 0; m[:, 3, 3] = 1
    return m


def fk_forward32_op_by_op(angles, bone_len, root):
    """fk_forward32 at the REFERENCE'S OP GRANULARITY (bench.py's cpu_baseline_faithful: the same arithmetic issued as the
    reference issues it, so that the CPU-vs-GPU ratio can be split into "fewer, fused ops" and "hardware"):
    33 per-joint matrix builds (_dh_matrix_slicewise), the arm chains re-multiplying the body prefix, 46 sequential bmm on
    cloned operands written back in place, cloned translation columns, one global-rotation bmm per chain and coordinate
    block, one slice assignment per output coordinate (R/models_Fk_GAN/forward_kinematics_DH_model.py:592-822).
    Bit-identical to fk_forward32 (tests/test_oracle_golden.py)."""
    N = angles.shape[0]
    rl, ll, bd, ra, la = _chain_tables(bone_len, angles.dtype)

    def build(tab, ang, n, lead=None):
        hm = torch.zeros((N, n, 4, 4), dtype=torch.float32)
        off = 0
        if lead is not None:
            hm[:, 0:9] = torch.clone(lead[:, 0:9])
            off = 9
        for i in range(n - off):
            hm[:, i + off] = _dh_matrix_slicewise(tab["alpha"][:, i], tab["a"][:, i], tab["d"][:, i], tab["theta"][:, i] + ang[:, i])
        return hm

    ll_hm = build(ll, angles[:, 5:10], 5)
    rl_hm = build(rl, angles[:, 0:5], 5)
    bd_hm = build(bd, angles[:, 10:23], 13)
    ra_hm = build(ra, angles[:, 23:28], 14, lead=bd_hm)              # (the prefix is copied BEFORE the body chain is multiplied out)
    la_hm = build(la, angles[:, 28:33], 14, lead=bd_hm)
    for hm in (ll_hm, rl_hm, bd_hm, ra_hm, la_hm):
        for i in range(hm.shape[1] - 1):
            hm[:, i + 1] = torch.bmm(torch.clone(hm[:, i]), torch.clone(hm[:, i + 1]))
    Rg = rotation_matrix(angles[:, 34], angles[:, 35], angles[:, 36])

    def rotated(hm):
        X, Y, Z = torch.clone(hm[:, :, 0, 3]), torch.clone(hm[:, :, 1, 3]), torch.clone(hm[:, :, 2, 3])
        p = torch.zeros((N, 3, hm.shape[1]), dtype=torch.float32)
        p[:, 0, :] = X; p[:, 1, :] = Y; p[:, 2, :] = Z
        q = torch.bmm(Rg, p)
        return torch.clone(q[:, 0, :]), torch.clone(q[:, 1, :]), torch.clone(q[:, 2, :])

    out = torch.zeros((N, 32, 3), dtype=torch.float32)
    slots = ((bd_hm, ((0, 0), (12, 3), (13, 6), (14, 12), (15, 12))), (rl_hm, ((1, 0), (2, 3), (3, 4))),
             (ll_hm, ((6, 0), (7, 3), (8, 4))), (la_hm, ((17, 9), (18, 12), (19, 13))), (ra_hm, ((25, 9), (26, 12), (27, 13))))
    for hm, pairs in slots:
        X, Y, Z = rotated(hm)
        for slot, j in pairs:
            out[:, slot, 0] = X[:, j]; out[:, slot, 1] = Y[:, j]; out[:, slot, 2] = Z[:, j]
    return out + root.reshape(-1, 1, 3)


def fk_forward16(angles, bone_len, root):
    """FK followed by the 32->16 joint gather (R/models_Fk_GAN/Fk_generator.py:259)."""
    return fk_forward32(angles, bone_len, root)[:, H36M_32_TO_16]


def fk_scalar_numpy(angles, bone_len, root):
    """One pose, float64 scalars -- restates the numpy branch
    R/models_Fk_GAN/forward_kinematics_DH_model.py:366-560 (used by init_Fk_DH_angle :824-858).
    Returns (32,3) float32."""
    angles = np.asarray(angles, dtype=np.float64)
    L = {n: float(bone_len[i]) for i, n in enumerate(BONE_NAMES)}

    def dh(alpha, a, d, theta):
        al, th = alpha / 180 * np.pi, theta / 180 * np.pi
        return np.array([[math.cos(th), -math.sin(th), 0, a],
                         [math.sin(th) * math.cos(al), math.cos(th) * math.cos(al), -math.sin(al), -math.sin(al) * d],
                         [math.sin(th) * math.sin(al), math.cos(th) * math.sin(al), math.cos(al), math.cos(al) * d],
                         [0, 0, 0, 1.0]])

    def chain(tab, ang, a_over=None, d_over=None, prefix=None):
        a, d = list(tab["a"]), list(tab["d"])
        for k, v in (a_over or {}).items():
            a[k] = v
        for k, v in (d_over or {}).items():
            d[k] = v
        ms = list(prefix or []) + [dh(tab["alpha"][i], a[i], d[i], tab["theta"][i] + ang[i]) for i in range(len(a))]
        return ms

    def cum(ms):
        out = [ms[0]]
        for m in ms[1:]:
            out.append(out[-1].dot(m))
        return np.array([[c[0, 3], c[1, 3], c[2, 3]] for c in out]).T      # (3,n)

    body_local = chain(BODY, angles[10:23], {12: L["neck"]}, {3: L["waist"], 6: L["thorax"]})
    rl = chain(RLEG, angles[0:5], {0: L["right_hip"], 3: L["right_big_leg"], 4: L["right_small_leg"]})
    ll = chain(LLEG, angles[5:10], {0: -L["left_hip"], 3: L["left_big_leg"], 4: L["left_small_leg"]})
    ra = chain(RARM, angles[23:28], {0: -L["right_shoulder"], 3: L["right_big_arm"], 4: L["right_small_arm"]},
               prefix=body_local[0:9])
    la = chain(LARM, angles[28:33], {0: L["left_shoulder"], 3: L["left_big_arm"], 4: L["left_small_arm"]},
               prefix=body_local[0:9])
    ax, ay, az = (angles[34:37] / 180 * np.pi)
    R1 = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    R2 = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    R3 = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    Rg = R1.dot(R2).dot(R3)
    bd_p, rl_p, ll_p, ra_p, la_p = (Rg.dot(cum(c)) for c in (body_local, rl, ll, ra, la))
    out = np.zeros((32, 3))
    out[0] = bd_p[:, 0]
    out[1], out[2], out[3] = rl_p[:, 0], rl_p[:, 3], rl_p[:, 4]
    out[6], out[7], out[8] = ll_p[:, 0], ll_p[:, 3], ll_p[:, 4]
    out[12], out[13], out[14], out[15] = bd_p[:, 3], bd_p[:, 6], bd_p[:, 12], bd_p[:, 12]
    out[17], out[18], out[19] = la_p[:, 9], la_p[:, 12], la_p[:, 13]
    out[25], out[26], out[27] = ra_p[:, 9], ra_p[:, 12], ra_p[:, 13]
    return (out + np.asarray(root, dtype=np.float64)).astype(np.float32)


TPOSE_BONE_LEN = [0.5, 0.5, 0.6, 0.6, 0.25, 0.25, 0.25, 0.2, 0.4, 0.4, 0.4, 0.4, 0.35, 0.35, 0.15]   # :840-854


# ----------------------------------------------------------------------------------------------
# a8 / a11: bone vectors and KCS features
# ----------------------------------------------------------------------------------------------
def bone_vectors(pose16):
    """(N,16,3) -> (N,15,3) child - parent.  R/models_Fk_GAN/special_operate.py:513-539."""
    p = torch.tensor([a for a, _ in BONE_PAIRS])
    c = torch.tensor([b for _, b in BONE_PAIRS])
    return pose16[:, c] - pose16[:, p]


def bone_lengths(pose16):
    """R/models_Fk_GAN/Fk_generator.py:107-111."""
    bv = bone_vectors(pose16.reshape(-1, 16, 3))
    return torch.sqrt(torch.sum(bv ** 2, dim=-1))


def kcs_features(pose16, with_lengths=True):
    """(N,16,3)|(N,48) -> (N,30): 15 cosines between adjacent bones then the 15 bone lengths
    (R/models_Fk_GAN/Fk_discriminator.py:36-146); with_lengths=False gives the video variant (:269-377)."""
    bv = bone_vectors(pose16.reshape(-1, 16, 3))
    bl = torch.sqrt(torch.sum(bv ** 2, dim=-1))
    i = torch.tensor([a for a, _ in KCS_PAIRS])
    j = torch.tensor([b for _, b in KCS_PAIRS])
    cos = torch.sum(bv[:, i] * bv[:, j], dim=-1) / (bl[:, i] * bl[:, j])
    return torch.cat([cos, bl], dim=-1) if with_lengths else cos


# ----------------------------------------------------------------------------------------------
# Dense layers.  precision: 'fp32' (reference arithmetic) or 'bf16' (emulates the build's choice:
# bf16-rounded operands, fp32 accumulate, activations stored as bf16 between layers); 'bf16_fused': the same with the one
# extra rounding point of the build's one-launch 3D-critic program (d3_forward).
# ----------------------------------------------------------------------------------------------
def _rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _linear(x, sd, key, precision):
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    if precision in ("bf16", "bf16_fused"):
        return F.linear(_rb(x), _rb(w)) + b
    return F.linear(x, w, b)


def _store(x, precision):
    # (autograd through the dtype round trip casts the COTANGENT to bf16 as well -- the gradient of a bf16 tensor is bf16 -- so a
    # backward / double backward through this forward rounds cotangents and tangents where the build's explicit training step
    # stores them in bf16 buffers)
    return _rb(x) if precision in ("bf16", "bf16_fused") else x


def resblock(x, sd, key, precision="fp32"):
    """relu(fc2(relu(fc1(x))) + x) -- R/models_Fk_GAN/special_operate.py:490-510."""
    h = _store(torch.relu(_linear(x, sd, key + ".fc1", precision)), precision)
    return _store(torch.relu(_linear(h, sd, key + ".fc2", precision) + x), precision)


def gen_trunk(z, sd, precision="fp32"):
    """z (B,128) -> head pre-activation (B, 35*R).  R/models_Fk_GAN/Fk_generator.py:115-119."""
    x = _store(torch.relu(_linear(_store(z, precision), sd, "preprocess.0", precision)), precision)
    for k in ("block1", "block2", "block3"):
        x = resblock(x, sd, k, precision)
    return _linear(x, sd, "deconv_out", precision)


def gen_tail_angles(head, use_preangle=True):
    """head (N,35) pre-activation -> (generator_angle (N,37) degrees, root (N,3)).
    R/models_Fk_GAN/Fk_generator.py:121-168."""
    t = torch.tanh(head[:, :-3])
    root = torch.tanh(head[:, -3:]) * 10.0
    N = head.shape[0]
    g = torch.zeros((N, 37), dtype=head.dtype)
    g[:, LIVE_SLOTS] = t[:, :31]
    if use_preangle:
        lo = torch.tensor(ANGLE_LO, dtype=head.dtype)
        hi = torch.tensor(ANGLE_HI, dtype=head.dtype)
        g = g * (hi - lo) / 2 + (hi + lo) / 2
    else:
        g = g * 180
    return g, root


def jitter_bone_len(bone_len, scaler):
    """len * (1 + s[pair]) -- R/models_Fk_GAN/Fk_generator.py:216-230.  scaler (N,8)."""
    cols = []
    for i in range(15):
        j = JITTER_COL[i]
        cols.append(bone_len[:, i] if j < 0 else bone_len[:, i] * (1 + scaler[:, j]))
    return torch.stack(cols, dim=1)


def gen_tail(head, bone_len, scaler, use_preangle=True, fk32=None):
    """head (N,35), bone_len (N,15), scaler (N,8) -> (fake (N,48), generator_angle (N,37)).  fk32: the FK restatement to use
    (default fk_forward32; bench.py's faithful CPU baseline passes fk_forward32_op_by_op)."""
    g, root = gen_tail_angles(head, use_preangle)
    bl = jitter_bone_len(bone_len, scaler)
    return (fk32 or fk_forward32)(g, bl, root)[:, H36M_32_TO_16].reshape(-1, 48), g


def generator_forward(z, sd, bone_len, scaler, use_preangle=True, frames=1, precision="fp32", fk32=None):
    """Fk_Generator.forward (frames=1, R/models_Fk_GAN/Fk_generator.py:114-261) /
    Video_Fk_Generator.forward (frames=R, :302-458; scaler (B,8) repeated over frames).
    Returns fake (B,48) or (B,R,48), plus the head pre-activation and the 37-angle tensor."""
    head = gen_trunk(z, sd, precision)
    B = z.shape[0]
    h = head.reshape(B * frames, 35)
    if frames > 1:
        scaler = scaler.reshape(B, 1, 8).repeat(1, frames, 1).reshape(B * frames, 8)
    fake, g = gen_tail(h, bone_len, scaler, use_preangle, fk32)
    if frames > 1:
        fake = fake.reshape(B, frames, 48)
    return fake, head, g


def d3_forward(x, sd, precision="fp32"):
    """Fk_3D_Discriminator.forward: root-relative pose (N,16,3)|(N,48) -> logit (N,1).
    R/models_Fk_GAN/Fk_discriminator.py:180-201."""
    x = x.reshape(-1, 48)
    k = _store(kcs_features(x), precision)
    k = _store(torch.relu(_linear(k, sd, "special_KCS_previous.0", precision)), precision)
    for n in ("special_KCS_block1", "special_KCS_block2", "special_KCS_block3"):
        k = resblock(k, sd, n, precision)
    p = _store(torch.relu(_linear(_store(x, precision), sd, "previous.0", precision)), precision)
    for n in ("block1", "block2", "block3"):
        p = resblock(p, sd, n, precision)
    if precision == "bf16_fused":
        # the build's FUSED bf16 programs (one launch per network; the layer-by-layer path is precision 'bf16') compute the merge layer in two halves: the KCS branch's share (with the bias) waits as
        # bf16 while the pose branch runs -- one more rounding point than the concatenated product (fused.py _d3_program)
        w, b = sd["merge_previous.0.weight"], sd["merge_previous.0.bias"]
        Dk = k.shape[-1]
        half = _rb(F.linear(_rb(k), _rb(w[:, :Dk])) + b)
        m = _store(torch.relu(half + F.linear(_rb(p), _rb(w[:, Dk:]))), precision)
    else:
        m = torch.cat([k, p], dim=-1)
        m = _store(torch.relu(_linear(m, sd, "merge_previous.0", precision)), precision)
    m = resblock(m, sd, "merge_block1", precision)
    return _linear(m, sd, "output", precision)


def d2_forward(x, sd, precision="fp32", slope=0.01):
    """Fk_2D_Discriminator.forward: (N,16,2)|(N,32) -> (N,1); LeakyReLU slope 0.01.
    R/models_Fk_GAN/Fk_discriminator.py:253-266."""
    x = _store(x.reshape(-1, 32), precision)
    lr = lambda t: F.leaky_relu(t, slope)
    d1 = _store(lr(_linear(x, sd, "pose_layer_1", precision)), precision)
    d2 = _store(lr(_linear(d1, sd, "pose_layer_2", precision)), precision)
    d3 = _store(lr(_linear(d2, sd, "pose_layer_3", precision) + d1), precision)
    d4 = _store(_linear(d3, sd, "pose_layer_4", precision), precision)
    dl = _store(lr(_linear(d4, sd, "layer_last", precision)), precision)
    return _linear(dl, sd, "layer_pred", precision)


def _frame_diff(x, frames, width):
    x = x.reshape(-1, frames, width)
    return (x[:, 1:] - x[:, :-1]).reshape(-1, (frames - 1) * width)


def motion_d3_forward(x, sd, frames, use_pos=True, use_diff=True, precision="fp32"):
    """Video_motion_Fk_3D_Discriminator.forward: (B*R,48) -> (B,1).
    R/models_Fk_GAN/Fk_discriminator.py:437-512."""
    x = x.reshape(-1, 48)
    kc = kcs_features(x, with_lengths=False).reshape(-1, frames * 15)

    def branch(inp, name):
        h = _store(torch.relu(_linear(_store(inp, precision), sd, name + "_previous.0", precision)), precision)
        for i in (1, 2, 3):
            h = resblock(h, sd, "%s_block%d" % (name, i), precision)
        return h

    outs = [branch(kc, "special_KCS"), branch(_frame_diff(kc, frames, 15), "diff_special_KCS")]
    if use_pos:
        outs.append(branch(x.reshape(-1, frames * 48), "pos_3d"))
    if use_diff:
        outs.append(branch(_frame_diff(x, frames, 48), "diff_pos_3d"))
    m = torch.cat(outs, dim=-1)
    m = _store(torch.relu(_linear(m, sd, "kcs_merge_previous.0", precision)), precision)
    m = resblock(m, sd, "kcs_merge_block1", precision)
    return _linear(m, sd, "kcs_output", precision)


def motion_d2_forward(x, sd, frames, precision="fp32"):
    """Video_motion_Fk_2D_Discriminator.forward: (B*R,32) -> (B,1).
    R/models_Fk_GAN/Fk_discriminator.py:548-587."""
    x = x.reshape(-1, 32)

    def branch(inp, name):
        h = _store(torch.relu(_linear(_store(inp, precision), sd, name + "_previous.0", precision)), precision)
        for i in (1, 2, 3):
            h = resblock(h, sd, "%s_block%d" % (name, i), precision)
        return h

    p = branch(x.reshape(-1, frames * 32), "pos_2d")
    r = branch(_frame_diff(x.reshape(-1, 16, 2)[:, 0, :], frames, 2), "root_diff_2d")
    m = torch.cat([p, r], dim=-1)
    m = _store(torch.relu(_linear(m, sd, "merge_previous.0", precision)), precision)
    m = resblock(m, sd, "merge_block1", precision)
    return _linear(m, sd, "merge_output", precision)


# ----------------------------------------------------------------------------------------------
# a14 / a15: WGAN-GP gradient penalty and critic step
# ----------------------------------------------------------------------------------------------
def gradient_penalty(d_fn, real, fake, alpha, lam=10.0):
    """lam * mean((||dD/dx_hat||_2 - 1)^2), x_hat = alpha*real + (1-alpha)*fake, alpha (B,1) injected.
    R/models_Fk_GAN/Fk_discriminator.py:205-231."""
    B = alpha.shape[0]
    real = real.reshape(B, -1)
    fake = fake.reshape(B, -1)
    xh = (alpha * real + (1 - alpha) * fake).detach().requires_grad_(True)
    out = d_fn(xh)
    g = torch.autograd.grad(out, xh, grad_outputs=torch.ones_like(out), create_graph=True)[0]
    return ((g.norm(2, dim=1) - 1) ** 2).mean() * lam


def critic_step(d_fn_sd, sd, real, fake, alpha, lam=10.0, lr=1e-4, betas=(0.5, 0.9), adam_state=None):
    """One train_Fk_discriminator call (R/models_Fk_GAN/model_fk_gan_train.py:177-230) on a parameter
    dict `sd` (name -> leaf tensor).  Returns dict(Wasserstein_D, D_cost, grads, new_params)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    fn = lambda x: d_fn_sd(x, params)
    d_real = fn(real).mean()
    d_fake = fn(fake).mean()
    gp = gradient_penalty(fn, real.detach(), fake.detach(), alpha, lam)
    # backward(mone) on D_real, backward(one) on D_fake, backward() on GP:
    total = -d_real + d_fake + gp
    grads = torch.autograd.grad(total, list(params.values()), allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(params.keys(), grads)}
    opt = torch.optim.Adam(list(params.values()), lr=lr, betas=betas)
    if adam_state is not None:
        opt.load_state_dict(adam_state)
    for k, p in params.items():
        p.grad = grads[k].clone()
    opt.step()
    return dict(Wasserstein_D=(d_real - d_fake).detach(), D_cost=(d_fake - d_real + gp).detach(),
                D_real=d_real.detach(), D_fake=d_fake.detach(), GP=gp.detach(),
                grads=grads, new_params={k: v.detach() for k, v in params.items()}, adam_state=opt.state_dict())


# ----------------------------------------------------------------------------------------------
# "next" row N1: camera transforms (world<->camera by quaternion, H36M projection, flip)
# ----------------------------------------------------------------------------------------------
def qrot(q, v):
    """R/common/quaternion.py:6-24."""
    qvec = q[..., 1:]
    uv = torch.cross(qvec, v, dim=-1)
    uuv = torch.cross(qvec, uv, dim=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


def world_to_camera(X, R, t):
    """R/common/camera.py:36-38 (qinverse then qrot of X - t).  X (N,16,3), R (1,4), t (1,3)."""
    Rt = torch.cat([R[..., :1], -R[..., 1:]], dim=-1)
    return qrot(Rt.expand(X.shape[:-1] + (4,)), X - t)


def camera_to_world(X, R, t):
    """R/common/camera.py:53-59.  X (N,16,3), R (N,4), t (N,3)."""
    R = R.reshape(-1, 1, 4).expand(-1, X.shape[1], -1)
    t = t.reshape(-1, 1, 3)
    return qrot(R, X) + t


def project_to_2d(X, cam):
    """H36M non-linear projection.  X (N,J,3) camera space, cam (N,9) = f(2) c(2) k(3) p(2).
    R/common/camera.py:62-94."""
    cam = cam.unsqueeze(1)
    f, c, k, p = cam[..., :2], cam[..., 2:4], cam[..., 4:7], cam[..., 7:9]
    XX = torch.clamp(X[..., :2] / X[..., 2:], min=-1, max=1)
    r2 = torch.sum(XX ** 2, dim=-1, keepdim=True)
    radial = 1 + torch.sum(k * torch.cat((r2, r2 ** 2, r2 ** 3), dim=-1), dim=-1, keepdim=True)
    tan = torch.sum(p * XX, dim=-1, keepdim=True)
    return f * (XX * (radial + tan) + p * r2) + c


def flip_lr(x):
    """negate x, swap left/right joints.  R/models_Fk_GAN/model_fk_gan_train.py:320-331."""
    y = x.detach().clone()
    y[:, :, 0] *= -1
    y[:, FLIP_LEFT + FLIP_RIGHT, :] = y[:, FLIP_RIGHT + FLIP_LEFT, :]
    return y


# ----------------------------------------------------------------------------------------------
# "next" row N3: bone-length swap of real poses (PoseAug bone algebra)
# ----------------------------------------------------------------------------------------------
PA_PARENT = [0, 1, 2, 0, 4, 5, 0, 7, 8, 8, 10, 11, 8, 13, 14]
PA_CHILD = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]


def random_bl_aug(x, new_len):
    """x (N,16,3), new_len (N,15) PoseAug bone order -> (N,16,3).  R/function_aug/dataloader_update.py:18-40 with the
    bone algebra of R/utils/gan_utils.py:56-138 (bone = parent - child, pose rebuilt as minus the path sums)."""
    root = x[:, :1, :] * 1.0
    x = x - x[:, :1, :]
    b = x[:, PA_PARENT] - x[:, PA_CHILD]
    unit = b / torch.norm(b, dim=2, keepdim=True)
    nb = unit * new_len.unsqueeze(2)
    out = [torch.zeros_like(x[:, 0])] * 16
    for k in range(15):
        out[PA_CHILD[k]] = out[PA_PARENT[k]] - nb[:, k]
    return torch.stack(out, dim=1) + root


# ----------------------------------------------------------------------------------------------
# a16 / a18: the epoch loops, restated on parameter dicts (name -> leaf tensor) with torch autograd + torch.optim.Adam.
# Random draws (noise, bone-length jitter, GP interpolation coefficients, camera choice) are INPUTS here, in the order
# the reference consumes them; tests/golden/make_golden_loops.py records them from the reference's own run.
# ----------------------------------------------------------------------------------------------
class Net:
    """parameter dict + forward function + Adam(1e-4, (0.5, 0.9)) (R/models_Fk_GAN/model_fk_gan_train.py:112-118)"""

    def __init__(self, sd, fwd):
        self.p = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        self.fwd = fwd
        self.opt = torch.optim.Adam(list(self.p.values()), lr=1e-4, betas=(0.5, 0.9))

    def __call__(self, x):
        return self.fwd(x, self.p)

    def zero_grad(self):
        for v in self.p.values():
            v.grad = None

    def grads(self):
        return {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in self.p.items()}

    def state(self):
        return {k: v.detach().clone() for k, v in self.p.items()}


def critic_step_net(net, real, fake, alpha, gp_rows, lam=10.0):
    """train_Fk_discriminator (R/models_Fk_GAN/model_fk_gan_train.py:177-230) on a Net; gp_rows = BATCH_SIZE handed to
    calc_gradient_penalty (args.batch_size * real_used_num).  Returns (Wasserstein_D, D_cost)."""
    net.zero_grad()
    d_real, d_fake = net(real).mean(), net(fake).mean()
    gp = gradient_penalty(net, real.detach().reshape(gp_rows, -1), fake.detach().reshape(gp_rows, -1), alpha, lam)
    (d_fake - d_real + gp).backward()
    net.opt.step()
    return (d_real - d_fake).detach(), (d_fake - d_real + gp).detach()


def gan_iteration(G, D3, D2, real_cam3d, cam_param, real2d, camera, noise, scaler, alphas, flip=True, g_step=None,
                  w3d=1.0, w2d=0.2, lam=10.0):
    """One pass of R/models_Fk_GAN/model_fk_gan_train.py:281-489.  G/D3/D2: Net objects (G.fwd = (z, p, bone_len,
    scaler) -> fake (B,48)); camera = (quat (1,4), trans (1,3), cam9 (B,9)); alphas: 4 (B,1) tensors in call order
    (D3, D3 flipped, D2, D2 flipped); g_step = dict(noise, scaler) on the iterations that step the generator.
    Returns dict(pos_3d_cam, pos_2d, W3, C3, W2, C2, G_cost, g_grads)."""
    quat, trans, cam9 = camera
    B = real_cam3d.shape[0]
    bl = bone_lengths(real_cam3d)
    real_w = camera_to_world(real_cam3d.reshape(-1, 16, 3), cam_param[:, 9:13], cam_param[:, 13:16])
    real_c = real_w - real_w[:, :1]
    with torch.no_grad():
        fake_w = G.fwd(noise, G.p, bl, scaler).reshape(-1, 16, 3)
    fake_c = fake_w - fake_w[:, :1]
    al = list(alphas)
    W3, C3 = critic_step_net(D3, real_c, fake_c, al.pop(0), B, lam)
    if flip:
        Wf, Cf = critic_step_net(D3, flip_lr(real_c), flip_lr(fake_c), al.pop(0), B, lam)
        W3, C3 = (W3 + Wf) / 2, (C3 + Cf) / 2
    pos_3d_cam = world_to_camera(fake_w, quat, trans)
    pos_2d = project_to_2d(pos_3d_cam, cam9)
    W2, C2 = critic_step_net(D2, real2d, pos_2d, al.pop(0), B, lam)
    if flip:
        Wf, Cf = critic_step_net(D2, flip_lr(real2d), flip_lr(pos_2d), al.pop(0), B, lam)
        W2, C2 = (W2 + Wf) / 2, (C2 + Cf) / 2
    out = dict(pos_3d_cam=pos_3d_cam, pos_2d=pos_2d, W3=W3, C3=C3, W2=W2, C2=C2, G_cost=None, g_grads=None)
    if g_step is not None:                                     # :415-484
        G.zero_grad()
        fw = G.fwd(g_step["noise"], G.p, bl, g_step["scaler"]).reshape(-1, 16, 3)
        f2d = project_to_2d(world_to_camera(fw, quat, trans), cam9)
        fc = fw - fw[:, :1]
        a3, a2 = D3(fc).mean(), D2(f2d).mean()
        if flip:                                               # flipped copies: value only (.detach().clone(), :455,:459)
            a3 = (a3 + D3(flip_lr(fc)).mean()) / 2
            a2 = (a2 + D2(flip_lr(f2d)).mean()) / 2
        gen_loss = a3 * w3d + a2 * w2d
        grads = torch.autograd.grad(-gen_loss, list(G.p.values()))
        for v, g in zip(G.p.values(), grads):
            v.grad = g
        out["g_grads"] = G.grads()
        out["d_states"] = dict(d3=D3.state(), d2=D2.state())
        G.opt.step()
        out["G_cost"] = (-gen_loss).detach()
    return out


def _rev(x, R, width):
    return torch.flip(x.reshape(-1, R, width), dims=[1])


def video_gan_iteration(G, D3, D2, M3, M2, R, real_cam3d, cam_param, real2d, camera, noise, scaler, alphas, flip=True,
                        playback=True, motion_on=True, g_step=None, w=(1.0, 0.2, 1.0, 1.0), lam=10.0):
    """One pass of R/models_Fk_GAN/video_GAN_fun.py:156-566.  real_cam3d (B,R,16,3), cam_param (B,16), real2d (B,R,16,2);
    camera cam9 has B*R rows; alphas in the reference's call order.  Critic-step conventions of the reference:
    3D motion critic steps use dis_mode='motion' (GP over B clips of R*48), 2D motion critic steps use the default mode
    (GP over B*R frames of 32, :341-346).  The G step views the 3D clip as (-1, R, 32) before the time flip (q6, :467,:521)."""
    quat, trans, cam9 = camera
    B = real_cam3d.shape[0]
    bl = bone_lengths(real_cam3d.reshape(-1, 16, 3))
    cR = cam_param[:, 9:13].unsqueeze(1).repeat(1, R, 1).reshape(-1, 4)
    cT = cam_param[:, 13:16].unsqueeze(1).repeat(1, R, 1).reshape(-1, 3)
    real_w = camera_to_world(real_cam3d.reshape(-1, 16, 3), cR, cT)
    real = (real_w - real_w[:, :1]).reshape(-1, 48)
    with torch.no_grad():
        fake_w = G.fwd(noise, G.p, bl, scaler).reshape(-1, 16, 3)
    fake = (fake_w - fake_w[:, :1]).reshape(-1, 48)
    al = list(alphas)
    avg = lambda a, b: tuple((x + y) / 2 for x, y in zip(a, b))
    out = {}

    def steps3(r, f):
        d = critic_step_net(D3, r, f, al.pop(0), B * R, lam)
        m = None
        if motion_on:
            m = critic_step_net(M3, r, f, al.pop(0), B, lam)
        if playback and motion_on:
            m = avg(m, critic_step_net(M3, _rev(r, R, 48), _rev(f, R, 48), al.pop(0), B, lam))
        return d, m

    out["d3"], out["m3"] = steps3(real, fake)
    if flip:
        fl = lambda x: flip_lr(x.reshape(-1, 16, 3)).reshape(-1, 48)
        d, m = steps3(fl(real), fl(fake))
        out["d3"] = avg(out["d3"], d)
        if motion_on:
            out["m3"] = avg(out["m3"], m)
    pos_3d_cam = world_to_camera(fake_w, quat, trans)
    pos_2d = project_to_2d(pos_3d_cam, cam9)
    r2 = real2d.reshape(-1, 16, 2)

    def steps2(r, f):
        d = critic_step_net(D2, r, f, al.pop(0), B * R, lam)
        m = None
        if motion_on:
            m = critic_step_net(M2, r, f, al.pop(0), B * R, lam)
        if playback and motion_on:
            m = avg(m, critic_step_net(M2, _rev(r, R, 32), _rev(f, R, 32), al.pop(0), B * R, lam))
        return d, m

    out["d2"], out["m2"] = steps2(r2, pos_2d)
    if flip:
        d, m = steps2(flip_lr(r2), flip_lr(pos_2d))
        out["d2"] = avg(out["d2"], d)
        if motion_on:
            out["m2"] = avg(out["m2"], m)
    assert not al, "unused GP coefficients"
    out.update(pos_3d_cam=pos_3d_cam.reshape(B, R, 16, 3), pos_2d=pos_2d.reshape(B, R, 16, 2), G_cost=None, g_grads=None)
    if g_step is not None:
        G.zero_grad()
        fw = G.fwd(g_step["noise"], G.p, bl, g_step["scaler"]).reshape(-1, 16, 3)
        f2d = project_to_2d(world_to_camera(fw, quat, trans), cam9)
        fc = fw - fw[:, :1]

        def terms(fc, f2d):
            a3, a2 = D3(fc).mean(), D2(f2d).mean()
            am3 = am2 = 0.0
            if motion_on:
                am3, am2 = M3(fc).mean(), M2(f2d).mean()
                if playback:
                    am3 = (am3 + M3(torch.flip(fc.reshape(-1, R, 32), dims=[1])).mean()) / 2      # q6
                    am2 = (am2 + M2(torch.flip(f2d.reshape(-1, R, 32), dims=[1])).mean()) / 2
            return [a3, a2, am3, am2]

        t = terms(fc, f2d)
        if flip:
            with torch.no_grad():
                tf = terms(flip_lr(fc), flip_lr(f2d))
            t = [(a + b) / 2 for a, b in zip(t, tf)]
        gen_loss = t[0] * w[0] + t[1] * w[1]
        if motion_on:
            gen_loss = gen_loss + t[2] * w[2] + t[3] * w[3]
        grads = torch.autograd.grad(-gen_loss, list(G.p.values()))
        for v, g in zip(G.p.values(), grads):
            v.grad = g
        out["g_grads"] = G.grads()
        G.opt.step()
        out["G_cost"] = (-gen_loss).detach()
    return out
