"""Tensor-level wrappers of the C-ABI (no autograd here): allocate outputs with torch, pass raw device pointers and
the current HIP stream to libdhaug.so.  Every function requires CUDA(HIP) tensors and raises otherwise."""
import ctypes
import os

import torch

from . import _lib

_vp = ctypes.c_void_p
BF16 = torch.bfloat16


def _stream():
    """raw handle of torch's current stream (an int: argtypes c_void_p converts it; no ctypes object per call)"""
    return torch.cuda.current_stream().cuda_stream or None


def _p(t):
    """raw device pointer (int) or None"""
    return None if t is None else t.data_ptr()


def _dev(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError("dhaug op `%s` needs a GPU tensor (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        t = t.to(dtype)
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _host3(x, n):
    """small host-side float arrays (camera parameters) -> ctypes float[n]"""
    if torch.is_tensor(x):
        x = x.detach().reshape(-1).cpu().tolist()
    x = [float(v) for v in (x.reshape(-1).tolist() if hasattr(x, "reshape") else x)]
    assert len(x) == n, (len(x), n)
    return (ctypes.c_float * n)(*x)


def ceil_to(v, m):
    return (v + m - 1) // m * m


# ---------------------------------------------------------------------------------------------- FK
def fk_forward(angles, bone_len, root, out_joints=16):
    a = _dev(angles, torch.float32, "fk_forward").reshape(-1, 37)
    b = _dev(bone_len, torch.float32, "fk_forward").reshape(-1, 15)
    r = None if root is None else _dev(root, torch.float32, "fk_forward").reshape(-1, 3)
    N = a.shape[0]
    assert b.shape[0] == N and (r is None or r.shape[0] == N), "FK inputs disagree on the number of poses"
    out = torch.empty((N, out_joints, 3), dtype=torch.float32, device=a.device)
    _lib.call("dhaug_fk_forward", _p(a), _p(b), _p(r), _p(out), N, out_joints, _stream())
    return out


def fk_backward(angles, bone_len, grad_out16):
    a = _dev(angles, torch.float32, "fk_backward").reshape(-1, 37)
    b = _dev(bone_len, torch.float32, "fk_backward").reshape(-1, 15)
    g = _dev(grad_out16, torch.float32, "fk_backward").reshape(-1, 48)
    N = a.shape[0]
    ga = torch.empty((N, 37), dtype=torch.float32, device=a.device)
    gb = torch.empty((N, 15), dtype=torch.float32, device=a.device)
    gr = torch.empty((N, 3), dtype=torch.float32, device=a.device)
    _lib.call("dhaug_fk_backward", _p(a), _p(b), _p(g), _p(ga), _p(gb), _p(gr), N, _stream())
    return ga, gb, gr


def gen_tail_forward(head, bone_len, scaler, use_preangle=True, want_angles=False):
    h = _dev(head, torch.float32, "gen_tail_forward").reshape(-1, 35)
    b = _dev(bone_len, torch.float32, "gen_tail_forward").reshape(-1, 15)
    s = None if scaler is None else _dev(scaler, torch.float32, "gen_tail_forward").reshape(-1, 8)
    N = h.shape[0]
    assert b.shape[0] == N and (s is None or s.shape[0] == N)
    fake = torch.empty((N, 16, 3), dtype=torch.float32, device=h.device)
    ang = torch.empty((N, 37), dtype=torch.float32, device=h.device) if want_angles else None
    _lib.call("dhaug_gen_tail_forward", _p(h), _p(b), _p(s), _p(fake), _p(ang), N, int(bool(use_preangle)), _stream())
    return fake, ang


def gen_tail_forward_critics(head, bone_len, scaler, use_preangle=True, camera=None, rng=None, want_scaler=False,
                             want_critic_inputs=True, inputs_bf16=False):
    """generator tail + what the critics consume in one launch: (fake (N,16,3), centered (N,48), kcs bf16 (N,32),
    proj2d (N,16,2) | None).  camera = (quat[4], trans[3], cam9[9]) host sequences.  rng = (seed, offset): draw the
    bone-length jitter in the kernel (scaler must be None); want_scaler appends the (N,8) draw to the result;
    want_critic_inputs=False skips the centred pose and the KCS operand (plain sampling with the in-kernel jitter);
    inputs_bf16: the centred pose and the projection leave as bf16 (the rounding the critics apply on load anyway)."""
    h = _dev(head, torch.float32, "gen_tail_forward_critics").reshape(-1, 35)
    b = _dev(bone_len, torch.float32, "gen_tail_forward_critics").reshape(-1, 15)
    s = None if scaler is None else _dev(scaler, torch.float32, "gen_tail_forward_critics").reshape(-1, 8)
    N = h.shape[0]
    assert b.shape[0] == N and (s is None or s.shape[0] == N)
    fake = torch.empty((N, 16, 3), dtype=torch.float32, device=h.device)
    idt = BF16 if inputs_bf16 else torch.float32
    xc = torch.empty((N, 48), dtype=idt, device=h.device) if want_critic_inputs else None
    kcs = torch.empty((N, 32), dtype=BF16, device=h.device) if want_critic_inputs else None
    p2 = q = t = c = None
    if camera is not None:
        p2 = torch.empty((N, 16, 2), dtype=idt, device=h.device)
        q, t, c = _host3(camera[0], 4), _host3(camera[1], 3), _host3(camera[2], 9)
    so = torch.empty((N, 8), dtype=torch.float32, device=h.device) if (rng is not None and want_scaler) else None
    _lib.call("dhaug_gen_tail_forward_critics", _p(h), _p(b), _p(s), _p(fake), _p(xc), _p(kcs), q, t, c, _p(p2),
              int(rng is not None), 0 if rng is None else int(rng[0]) & (2 ** 64 - 1), 0 if rng is None else int(rng[1]), _p(so),
              N, int(bool(use_preangle)), int(bool(inputs_bf16)), _stream())
    if want_scaler:
        return fake, xc, kcs, p2, so
    return fake, xc, kcs, p2


def gen_tail_backward(head, bone_len, scaler, grad_fake16, use_preangle=True):
    h = _dev(head, torch.float32, "gen_tail_backward").reshape(-1, 35)
    b = _dev(bone_len, torch.float32, "gen_tail_backward").reshape(-1, 15)
    s = None if scaler is None else _dev(scaler, torch.float32, "gen_tail_backward").reshape(-1, 8)
    g = _dev(grad_fake16, torch.float32, "gen_tail_backward").reshape(-1, 48)
    N = h.shape[0]
    gh = torch.empty((N, 35), dtype=torch.float32, device=h.device)
    _lib.call("dhaug_gen_tail_backward", _p(h), _p(b), _p(s), _p(g), _p(gh), N, int(bool(use_preangle)), _stream())
    return gh


# ------------------------------------------------------------------------------------- pose features
def bone_length(pose16):
    x = _dev(pose16, torch.float32, "bone_length").reshape(-1, 48)
    out = torch.empty((x.shape[0], 15), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_bone_length", _p(x), _p(out), x.shape[0], _stream())
    return out


def kcs_forward(pose16, with_lengths=True, f32=True, bf16_ld=0):
    x = _dev(pose16, torch.float32, "kcs_forward").reshape(-1, 48)
    N, W = x.shape[0], (30 if with_lengths else 15)
    of = torch.empty((N, W), dtype=torch.float32, device=x.device) if f32 else None
    ob = torch.empty((N, bf16_ld), dtype=BF16, device=x.device) if bf16_ld else None
    _lib.call("dhaug_kcs_forward", _p(x), _p(of), _p(ob), bf16_ld, N, int(with_lengths), _stream())
    return of, ob


def center_kcs_forward(pose16, bf16_ld=32, with_lengths=True):
    """root-relative pose (N,48) fp32 and the bf16 KCS operand of the 3D critic in one pass over the pose"""
    x = _dev(pose16, torch.float32, "center_kcs_forward").reshape(-1, 48)
    N = x.shape[0]
    xc = torch.empty((N, 48), dtype=torch.float32, device=x.device)
    ob = torch.empty((N, bf16_ld), dtype=BF16, device=x.device)
    _lib.call("dhaug_center_kcs_forward", _p(x), _p(xc), _p(ob), bf16_ld, N, int(with_lengths), _stream())
    return xc, ob


def kcs_backward(pose16, grad_feat, with_lengths=True):
    x = _dev(pose16, torch.float32, "kcs_backward").reshape(-1, 48)
    g = _dev(grad_feat, torch.float32, "kcs_backward").reshape(x.shape[0], 30 if with_lengths else 15)
    out = torch.empty((x.shape[0], 48), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_kcs_backward", _p(x), _p(g), _p(out), x.shape[0], int(with_lengths), _stream())
    return out


def kcs_jvp(pose16, tangent, with_lengths=True):
    x = _dev(pose16, torch.float32, "kcs_jvp").reshape(-1, 48)
    t = _dev(tangent, torch.float32, "kcs_jvp").reshape(-1, 48)
    out = torch.empty((x.shape[0], 30 if with_lengths else 15), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_kcs_jvp", _p(x), _p(t), _p(out), x.shape[0], int(with_lengths), _stream())
    return out


# -------------------------------------------------------------------------------------------- camera
def world_to_camera_project(pose16, quat, trans, cam9, want3d=True, want2d=True):
    x = _dev(pose16, torch.float32, "world_to_camera_project").reshape(-1, 48)
    N = x.shape[0]
    c3 = torch.empty((N, 16, 3), dtype=torch.float32, device=x.device) if want3d else None
    p2 = torch.empty((N, 16, 2), dtype=torch.float32, device=x.device) if want2d else None
    _lib.call("dhaug_world_to_camera_project", _p(x), _host3(quat, 4), _host3(trans, 3),
              None if cam9 is None else _host3(cam9, 9), _p(c3), _p(p2), N, _stream())
    return c3, p2


def world_to_camera_project_backward(pose16, quat, trans, cam9, grad3d, grad2d):
    x = _dev(pose16, torch.float32, "w2c_backward").reshape(-1, 48)
    g3 = None if grad3d is None else _dev(grad3d, torch.float32, "w2c_backward").reshape(-1, 48)
    g2 = None if grad2d is None else _dev(grad2d, torch.float32, "w2c_backward").reshape(-1, 32)
    out = torch.empty((x.shape[0], 16, 3), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_world_to_camera_project_backward", _p(x), _host3(quat, 4), _host3(trans, 3),
              None if cam9 is None else _host3(cam9, 9), _p(g3), _p(g2), _p(out), x.shape[0], _stream())
    return out


def camera_to_world(cam3d, quat, trans):
    x = _dev(cam3d, torch.float32, "camera_to_world").reshape(-1, 48)
    q = _dev(quat, torch.float32, "camera_to_world").reshape(-1, 4)
    t = _dev(trans, torch.float32, "camera_to_world").reshape(-1, 3)
    assert q.shape[0] == x.shape[0] and t.shape[0] == x.shape[0]
    out = torch.empty((x.shape[0], 16, 3), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_camera_to_world", _p(x), _p(q), _p(t), _p(out), x.shape[0], _stream())
    return out


def bone_length_swap(pose16, new_len):
    x = _dev(pose16, torch.float32, "bone_length_swap").reshape(-1, 48)
    l = _dev(new_len, torch.float32, "bone_length_swap").reshape(-1, 15)
    assert l.shape[0] == x.shape[0]
    out = torch.empty((x.shape[0], 16, 3), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_bone_length_swap", _p(x), _p(l), _p(out), x.shape[0], _stream())
    return out


def project_to_2d(cam3d, cam9):
    x = _dev(cam3d, torch.float32, "project_to_2d").reshape(-1, 48)
    c = _dev(cam9, torch.float32, "project_to_2d")[:, :9].contiguous()
    assert c.shape[0] == x.shape[0]
    out = torch.empty((x.shape[0], 16, 2), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_project_to_2d", _p(x), _p(c), _p(out), x.shape[0], _stream())
    return out


def center_flip(x, center, flip, adjoint=False):
    C = x.shape[-1]
    v = _dev(x, torch.float32, "center_flip").reshape(-1, 16 * C)
    out = torch.empty_like(v)
    _lib.call("dhaug_center_flip_backward" if adjoint else "dhaug_center_flip", _p(v), _p(out), v.shape[0], C,
              int(bool(center)), int(bool(flip)), _stream())
    return out.reshape(-1, 16, C)


# ---------------------------------------------------------------------------------------------- GEMM
def cast_pad_bf16(src, pad_cols=None, out=None):
    """fp32 (rows, cols) -> bf16 (rows, pad_cols), zero-padded; out: write into these rows of a larger buffer"""
    s = _dev(src, torch.float32, "cast_pad_bf16")
    s = s.reshape(-1, s.shape[-1])
    rows, cols = s.shape
    pad_cols = ceil_to(cols, 16) if pad_cols is None else pad_cols
    dst = torch.empty((rows, pad_cols), dtype=BF16, device=s.device) if out is None else out
    assert dst.dtype == BF16 and dst.shape == (rows, pad_cols) and dst.stride(1) == 1
    _lib.call("dhaug_cast_pad_bf16", _p(s), cols, _p(dst), dst.stride(0), rows, cols, pad_cols, _stream())
    return dst


def cast_transpose_bf16(src, pad_cols=None):
    s = _dev(src, torch.float32, "cast_transpose_bf16")
    rows, cols = s.shape
    pad_cols = ceil_to(rows, 16) if pad_cols is None else pad_cols
    dst = torch.empty((cols, pad_cols), dtype=BF16, device=s.device)
    _lib.call("dhaug_cast_transpose_bf16", _p(s), cols, _p(dst), pad_cols, rows, cols, pad_cols, _stream())
    return dst


def split_bf16(src, mode, terms, pad_cols=None):
    """fp32 (rows, cols) -> bf16 (rows, terms*pad): see dhaug_split_bf16 (mode 0 activation side, 1 weight side)."""
    if src.is_cuda and src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1 and src.stride(0) >= src.shape[1]:
        # a column block of a wider buffer is read where it lies (no contiguous copy, no realignment: the kernel's vector
        # path tests the base address and the row pitch itself and falls back to element loads)
        s, ld = src, src.stride(0)
    else:
        s = _dev(src, torch.float32, "split_bf16")
        s = s.reshape(-1, s.shape[-1])
        ld = s.shape[1]
    rows, cols = s.shape
    pad_cols = ceil_to(cols, 16) if pad_cols is None else pad_cols
    dst = torch.empty((rows, (3 if mode == 2 else terms) * pad_cols), dtype=BF16, device=s.device)
    _lib.call("dhaug_split_bf16", _p(s), ld, _p(dst), rows, cols, pad_cols, mode, terms, _stream())
    return dst


def gemm_planes_ok(N, kp, bias=None, res_f32=None, dmask_f32=None, out=None, six=False):
    """shapes dhaug_gemm_bf16x6_planes takes (the ping-pong tiles; a power-of-two piece width, or -- six: the ordinary six-segment operand,
    x_order 2 -- any width with 6 kp >= 128)"""
    al = lambda t: t is None or (t.data_ptr() % 16 == 0)
    row = lambda t: t is None or (t.stride(1) == 1 and t.stride(0) % 4 == 0)
    wide = (6 * kp >= 128 and kp % 8 == 0) if six else (kp >= 64 and (kp & (kp - 1)) == 0)
    return (N % 8 == 0 and wide and al(bias) and al(res_f32) and al(dmask_f32) and al(out) and row(res_f32) and row(dmask_f32) and row(out))


def gemm_nt_planes(A3, B6, N, kp, bias=None, res_f32=None, act=0, slope=0.0, dmask_f32=None, dmask_act=0, dmask_slope=0.0, out=None, x_order=0,
                   planes_out=False):
    """fp32 (M, N) = act(x W^T + bias + res_f32), masked by dmask_f32, in the bf16x6 arithmetic with the activation side as planes:
    A3 = split_bf16(x, 2, 6, kp) = [hi|mid|lo], B6 = split_bf16(W, 1, 6, kp) (dhaug_gemm_bf16x6_planes: bit-identical to gemm_nt on the
    mode 0 split); x_order 1: the planes stand for the mode 1 operand, B6 = split_bf16(W, 0, 6, kp)."""
    assert A3.dtype == BF16 and B6.dtype == BF16 and A3.is_cuda and B6.is_cuda and A3.stride(1) == 1 and B6.stride(1) == 1
    M = A3.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A3.device)
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[0] == M
    if bias is not None:
        bias = _dev(bias, torch.float32, "gemm_nt_planes")
    # planes_out: the result once more as split_bf16(out, 2, 6, N) would make it (N a multiple of 8), written by the GEMM's epilogue
    cp = torch.empty((M, 3 * N), dtype=BF16, device=A3.device) if planes_out else None
    _lib.call("dhaug_gemm_bf16x6_planes", _p(A3), A3.stride(0), _p(B6), B6.stride(0), _p(bias), _p(res_f32),
              0 if res_f32 is None else res_f32.stride(0), _p(dmask_f32), 0 if dmask_f32 is None else dmask_f32.stride(0), int(dmask_act),
              float(dmask_slope), _p(out), out.stride(0), _p(cp), 0 if cp is None else cp.stride(0), M, N, kp, int(x_order), act, float(slope),
              _stream())
    return (out, cp) if planes_out else out


def split_f16(src, mode, pad_cols=None):
    """fp32 (rows, cols) -> (rows, 3 * pad_cols) IEEE-half pieces of x = hi + lo: mode 0 [hi|hi|lo] (activation side), mode 1 [hi|lo|hi]
    (weight side) -- the operands of gemm_nt_f16x3.  |x| < 65 504."""
    s = _dev(src, torch.float32, "split_f16") if src.is_contiguous() or not (src.dim() == 2 and src.stride(1) == 1) else src
    assert s.dtype == torch.float32 and s.is_cuda and s.dim() == 2 and s.stride(1) == 1
    rows, cols = s.shape
    pad_cols = ceil_to(cols, 16) if pad_cols is None else pad_cols
    dst = torch.empty((rows, (2 if mode == 2 else 3) * pad_cols), dtype=torch.float16, device=s.device)
    _lib.call("dhaug_split_f16", _p(s), s.stride(0), _p(dst), rows, cols, pad_cols, mode, _stream())
    return dst


def gemm_nt_f16x3_planes(A, B, N, kp, a_planes, bias=None, res_f32=None, act=0, slope=0.0, planes_kp=0, out=None):
    """gemm_nt_f16x3 with the split left out of a chain of layers (dhaug_gemm_f16x3_planes): A = split_f16(x, 2, kp) = [hi|lo] (a_planes;
    kp = 64 * 2^j) or the mode 0 operand; B = split_f16(W, 1, kp); planes_kp > 0: returns (out, planes) with planes = split_f16(out, 2,
    planes_kp) written by the GEMM's epilogue."""
    assert A.dtype == torch.float16 and B.dtype == torch.float16 and A.is_cuda and B.is_cuda and A.stride(1) == 1
    M = A.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1      # (may be a column block of a wider buffer)
    cp = torch.empty((M, 2 * planes_kp), dtype=torch.float16, device=A.device) if planes_kp else None
    if bias is not None:
        bias = _dev(bias, torch.float32, "gemm_nt_f16x3_planes")
    _lib.call("dhaug_gemm_f16x3_planes", _p(A), A.stride(0), int(bool(a_planes)), _p(B), B.stride(0), _p(bias), _p(res_f32),
              0 if res_f32 is None else res_f32.stride(0), _p(out), out.stride(0), _p(cp), 0 if cp is None else cp.stride(0), int(planes_kp), M, N, kp,
              act, float(slope), _stream())
    return (out, cp) if planes_kp else out


def gemm_f16x3_ok(N, K3, bias=None, res_f32=None):
    """shapes dhaug_gemm_f16x3 takes (the ping-pong tiles: csrc/dhaug_gemm_p8.hip)"""
    al = lambda t: t is None or (t.data_ptr() % 16 == 0)
    return (N % 8 == 0 and K3 >= 128 and K3 % 16 == 0 and al(bias) and al(res_f32)
            and (res_f32 is None or (res_f32.stride(1) == 1 and res_f32.stride(0) % 4 == 0)))


def gemm_nt_f16x3(A, B, N, K3, bias=None, res_f32=None, act=0, slope=0.0, out=None):
    """fp32 (M, N) = act(A B^T + bias + res_f32) on IEEE-half operands: A = split_f16(x, 0), B = split_f16(W, 1), K3 = 3 * padded width
    (dhaug_gemm_f16x3: the "f16x3" arithmetic as a layer GEMM)."""
    assert A.dtype == torch.float16 and B.dtype == torch.float16 and A.is_cuda and B.is_cuda
    M = A.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[0] == M
    if bias is not None:
        bias = _dev(bias, torch.float32, "gemm_nt_f16x3")
    _lib.call("dhaug_gemm_f16x3", _p(A), A.stride(0), _p(B), B.stride(0), _p(bias), _p(res_f32), 0 if res_f32 is None else res_f32.stride(0),
              _p(out), out.stride(0), M, N, K3, act, float(slope), _stream())
    return out


def gemm_nt(A, B, N, K, bias=None, res_bf16=None, res_f32=None, act=0, slope=0.0, out_bf16=False, n_pad=0,
            out_f32=False, lda=None, ldb=None, c_bf16=None, c_f32=None):
    """C[M,N] = act(A[M,K] B[N,K]^T + bias + residual).  A, B bf16 (row strides lda/ldb default to their widths).
    Returns (c_bf16 (M, max(n_pad, ceil8(N))) | None, c_f32 (M,N) | None).  c_bf16 / c_f32: write into these (row-strided
    views allowed: a column block of a wider buffer) instead of allocating; columns [N, n_pad) of c_bf16 are zeroed."""
    assert A.dtype == BF16 and B.dtype == BF16 and A.is_cuda and B.is_cuda
    M = A.shape[0]
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    cb, cf = c_bf16, c_f32
    ldcb = 0
    if cb is not None:
        assert cb.dtype == BF16 and cb.stride(1) == 1 and cb.shape[0] == M
        ldcb, n_pad = cb.stride(0), min(max(n_pad, N), cb.shape[1])       # never past the view's own columns
    elif out_bf16:
        ldcb = max(n_pad, ceil_to(N, 8))
        cb = torch.empty((M, ldcb), dtype=BF16, device=A.device)
        n_pad = ldcb
    if cf is not None:
        assert cf.dtype == torch.float32 and cf.stride(1) == 1 and cf.shape[0] == M
    elif out_f32:
        cf = torch.empty((M, N), dtype=torch.float32, device=A.device)
    if bias is not None:
        bias = _dev(bias, torch.float32, "gemm_nt")
    _lib.call("dhaug_gemm_bf16", _p(A), lda, _p(B), ldb, _p(bias), _p(res_bf16),
              0 if res_bf16 is None else res_bf16.stride(0), _p(res_f32), 0 if res_f32 is None else res_f32.stride(0),
              _p(cb), ldcb, n_pad, _p(cf), N if cf is None else cf.stride(0), M, N, K, act, float(slope), _stream())
    return cb, cf


def gemm_nt_dmask(A, B, N, K, dmask, dmask_act, dmask_slope=0.0, res_bf16=None, out=None):
    """(A[M,K] B[N,K]^T + res) * act'(dmask): input gradient of a layer + activation backward of its producer in one launch;
    bf16 (M, ceil16 N) with zero pad columns, or `out` (a row-strided view: no columns beyond its own are touched)."""
    assert A.dtype == BF16 and B.dtype == BF16 and dmask.dtype == BF16
    M = A.shape[0]
    if out is None:
        out = torch.empty((M, ceil_to(N, 16)), dtype=BF16, device=A.device)
    assert out.dtype == BF16 and out.stride(1) == 1 and out.shape[0] == M and out.shape[1] >= N
    cols = getattr(dmask, "_dhaug_bits_cols", None)
    if (cols is not None and DBITS and N == 256 * len(cols) and len(cols) <= 2 and K in (16, 32, 48, 64, 112, 128, 256) and res_bf16 is None
            and dmask_act != 0 and M > 0 and A.stride(0) % 8 == 0 and out.stride(0) % 8 == 0):
        # a wide output whose 256-column blocks carry their own sign bits (the cotangent of the 3D critic's concatenation)
        assert all(c.device == A.device for c in cols), "sign bits must live on the operands' device"
        _lib.call("dhaug_gemm_bf16_dbits_wide", _p(A), A.stride(0), _p(B), B.stride(0), _p(cols[0]), _p(cols[1]) if len(cols) > 1 else 0,
                  dmask_act, float(dmask_slope), _p(out), out.stride(0), M, N, K, _stream())
        return out
    bits = getattr(dmask, "_dhaug_bits", None)
    assert bits is None or bits.device == A.device, "sign bits must live on the operands' device"
    if (bits is not None and DBITS and N == 256 and K in (16, 32, 48, 64, 112, 128) and res_bf16 is None and dmask_act != 0
            and M > 0 and M % 32 == 0 and A.stride(0) % 8 == 0 and out.stride(0) % 8 == 0):
        # a 256-wide layer behind a NARROW one (the tangent through a branch's first layer): the same mask bits, K < 256
        _lib.call("dhaug_gemm_bf16_dbits_wide", _p(A), A.stride(0), _p(B), B.stride(0), _p(bits), 0, dmask_act, float(dmask_slope),
                  _p(out), out.stride(0), M, N, K, _stream())
        return out
    if (bits is not None and DBITS and N == 256 and K == 256 and M % 32 == 0 and M > 0 and dmask_act != 0
            and A.stride(0) % 8 == 0 and out.stride(0) % 8 == 0):
        # the mask as the sign-bit array its forward-with-save layer left beside the image: 32 bytes per row instead of 512
        _lib.call("dhaug_gemm_bf16_dbits", _p(A), A.stride(0), _p(B), B.stride(0), _p(res_bf16),
                  0 if res_bf16 is None else res_bf16.stride(0), _p(bits), dmask_act, float(dmask_slope), _p(out), out.stride(0), M,
                  _stream())
        return out
    _lib.call("dhaug_gemm_bf16_dmask_pad", _p(A), A.stride(0), _p(B), B.stride(0), _p(res_bf16),
              0 if res_bf16 is None else res_bf16.stride(0), _p(dmask), dmask.stride(0), dmask_act, float(dmask_slope),
              _p(out), out.stride(0), min(out.shape[1], ceil_to(N, 16)), M, N, K, _stream())
    return out


NT_GROUP_MAX = 8
# the top of a branch critic (merge layer, merge block, logit layer) as one launch per sweep (dhaug_critic_top_*): DHAUG_NO_TOP_FUSED=1
# runs the separate launches
TOP_FUSED = os.environ.get("DHAUG_NO_TOP_FUSED") is None


def gemm_nt_group(members):
    """Up to NT_GROUP_MAX independent GEMMs of one shape as ONE launch (dhaug_gemm_bf16_group).  members: dicts with A, B (bf16
    operands), N, K and optionally bias, res_bf16, act, slope, dmask, dmask_act, dmask_slope, out (bf16 (M, >= N) view), n_pad.
    Each computes act(A B^T + bias + res) * dmask_act'(dmask) like gemm_nt / gemm_nt_dmask; returns the outputs."""
    assert 1 <= len(members) <= NT_GROUP_MAX
    arr = (_lib.GemmDesc * len(members))()
    outs, keep = [], []
    for d, m in zip(arr, members):
        A, B, N, K = m["A"], m["B"], m["N"], m["K"]
        assert A.dtype == BF16 and B.dtype == BF16
        M = A.shape[0]
        out = m.get("out")
        n_pad = m.get("n_pad", ceil_to(N, 16))
        if out is None:
            out = torch.empty((M, max(n_pad, ceil_to(N, 8))), dtype=BF16, device=A.device)
        assert out.dtype == BF16 and out.stride(1) == 1 and out.shape[0] == M
        n_pad = min(max(n_pad, N), out.shape[1])
        res, dm, bias = m.get("res_bf16"), m.get("dmask"), m.get("bias")
        d.A, d.lda, d.B, d.ldb = _p(A), A.stride(0), _p(B), B.stride(0)
        d.bias = _p(bias)
        d.residual, d.ld_res = _p(res), 0 if res is None else res.stride(0)
        d.residual_f32, d.ld_res_f32 = None, 0
        d.c_bf16, d.ldc_bf16, d.n_pad_zero = _p(out), out.stride(0), n_pad
        d.c_f32, d.ldc_f32 = None, 0
        d.M, d.N, d.K = M, N, K
        d.act, d.slope = int(m.get("act", 0)), float(m.get("slope", 0.0))
        d.dmask, d.ld_dmask = _p(dm), 0 if dm is None else dm.stride(0)
        d.dmask_act, d.dmask_slope = int(m.get("dmask_act", 0)) if dm is not None else 0, float(m.get("dmask_slope", 0.0))
        outs.append(out)
        keep.append((A, B, res, dm, bias))
    _lib.call("dhaug_gemm_bf16_group", arr, len(members), _stream())
    return outs


def gemm_nt_dmask_f32(A, B, N, K, dmask, dmask_act, dmask_slope=0.0, res_f32=None, out=None):
    """(A[M,K] B[N,K]^T + res) * act'(dmask) in the split-operand arithmetic: fp32 result / residual / mask (M, N), A and B the
    operands split_bf16 makes (K = terms * padded width)."""
    assert A.dtype == BF16 and B.dtype == BF16 and dmask.dtype == torch.float32 and dmask.stride(1) == 1
    M = A.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[0] == M and out.shape[1] >= N
    assert res_f32 is None or (res_f32.dtype == torch.float32 and res_f32.stride(1) == 1)
    _lib.call("dhaug_gemm_bf16_dmask_f32", _p(A), A.stride(0), _p(B), B.stride(0), _p(res_f32), 0 if res_f32 is None else res_f32.stride(0),
              _p(dmask), dmask.stride(0), dmask_act, float(dmask_slope), _p(out), out.stride(0), M, N, K, _stream())
    return out


DBITS = os.environ.get("DHAUG_NO_DBITS") is None          # consume sign-bit masks where a saved activation carries one
BLOCK2 = os.environ.get("DHAUG_NO_BLOCK2") is None        # the two layers of a residual block's backward / tangent step in one launch


def block2_ok(x, mask1, mask2, M):
    """can dhaug_gemm_block2_bf16 take this pair of 256 -> 256 layers?  (bf16 rows, whole 32-row tiles, both masks as sign bits)"""
    return (BLOCK2 and DBITS and M > 0 and M % 32 == 0 and x.dtype == BF16 and x.shape[1] >= 256 and x.stride(0) % 8 == 0
            and getattr(mask1, "_dhaug_bits", None) is not None and getattr(mask2, "_dhaug_bits", None) is not None)


def gemm_block2(x, B1, B2, mask1, mask2, act, slope=0.0, out1=None, out2=None):
    """y1 = (x B1^T) * act'(mask1); y2 = (y1 B2^T + x) * act'(mask2) -- x, y1, y2 bf16 (M, >= 256); B1, B2 bf16 (256, >= 256);
    the masks by their sign-bit arrays.  Returns (y1, y2)."""
    M = x.shape[0]
    if out1 is None:
        out1 = torch.empty((M, 256), dtype=BF16, device=x.device)
    if out2 is None:
        out2 = torch.empty((M, 256), dtype=BF16, device=x.device)
    assert out1.dtype == BF16 and out2.dtype == BF16 and out1.shape[0] == M and out2.shape[0] == M
    assert mask1._dhaug_bits.device == x.device and mask2._dhaug_bits.device == x.device, "sign bits must live on the operands' device"
    _lib.call("dhaug_gemm_block2_bf16", _p(x), x.stride(0), _p(B1), B1.stride(0), _p(B2), B2.stride(0), _p(mask1._dhaug_bits),
              _p(mask2._dhaug_bits), act, float(slope), _p(out1), out1.stride(0), _p(out2), out2.stride(0), M, _stream())
    return out1, out2


def gemm_block2_stack(x, blocks, act, slope=0.0):
    """a chain of gemm_block2 steps in one launch: blocks = [(B1, B2, mask1, mask2, out1 | None, out2 | None)], block i + 1 takes
    block i's y2 as its x.  Returns [(y1, y2)]."""
    M = x.shape[0]
    assert 1 <= len(blocks) <= _lib.BLOCK2_MAX and x.dtype == BF16
    arr = (_lib.Block2 * len(blocks))()
    outs = []
    for d, (B1, B2, mask1, mask2, out1, out2) in zip(arr, blocks):
        if out1 is None:
            out1 = torch.empty((M, 256), dtype=BF16, device=x.device)
        if out2 is None:
            out2 = torch.empty((M, 256), dtype=BF16, device=x.device)
        assert out1.dtype == BF16 and out2.dtype == BF16 and out1.shape[0] == M and out2.shape[0] == M
        assert mask1._dhaug_bits.device == x.device and mask2._dhaug_bits.device == x.device, "sign bits must live on the operands' device"
        d.W1, d.ldw1, d.W2, d.ldw2 = _p(B1), B1.stride(0), _p(B2), B2.stride(0)
        d.bits1, d.bits2 = _p(mask1._dhaug_bits), _p(mask2._dhaug_bits)
        d.Y1, d.ldy1, d.Y2, d.ldy2 = _p(out1), out1.stride(0), _p(out2), out2.stride(0)
        outs.append((out1, out2))
    _lib.call("dhaug_gemm_block2_stack_bf16", _p(x), x.stride(0), arr, len(blocks), act, float(slope), M, _stream())
    return outs


def tail_rows(t, r0):
    """t[r0:], keeping the sign-bit array of a saved activation attached (rows r0.. start at a 32-row tile)"""
    v = t[r0:]
    bits = getattr(t, "_dhaug_bits", None)
    if bits is not None and r0 % 32 == 0:
        v._dhaug_bits = bits[(r0 // 32) * 256:]
    return v


TN256 = os.environ.get("DHAUG_NO_TN256") is None
_TN_WS = {}


def _tn_group_workspace(dev):
    """DHAUG_TN_GROUP_WORKSPACE_FLOATS fp32 values per device: the per-workgroup partial results of dhaug_gemm_tn_group_bf16
    (any content; calls are ordered on the stream)"""
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture the buffer must belong to THAT graph's pool (a cached one would be memory of whichever
        # capture came first, gone with it): the pool hands the same block out again after each use
        return torch.empty(_lib.TN_GROUP_WORKSPACE_FLOATS, dtype=torch.float32, device=dev)
    k = (dev.type, dev.index, torch.cuda.current_stream().cuda_stream)      # (per stream: concurrent critic steps)
    if k not in _TN_WS:
        _TN_WS[k] = torch.empty(_lib.TN_GROUP_WORKSPACE_FLOATS, dtype=torch.float32, device=dev)
    return _TN_WS[k]


class workgroup_cap:
    """with workgroup_cap(n): the persistent launches issued inside take at most n workgroups (dhaug_set_workgroup_cap)"""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.old = _lib.lib().dhaug_set_workgroup_cap(self.n) if self.n else None
        return self

    def __exit__(self, *exc):
        if self.old is not None:
            _lib.lib().dhaug_set_workgroup_cap(self.old)


class nan_propagation:
    """with nan_propagation(True): the fused INFERENCE programs launched inside apply ReLU as max(v, v * 0) -- a NaN / inf input
    row reaches its logit, NaN weights reach every logit, as in the reference (dhaug_set_nan_propagation; process-wide default:
    DHAUG_NAN_PROPAGATION=1 in the environment).  The training iterations (gan_iteration, video_gan_iteration) always run inside it.
    The flag is a process-wide value read when a kernel is LAUNCHED: a hipGraph keeps the setting it was CAPTURED with, so wrapping
    the replay of an already captured graph changes nothing -- wrap the capture (or the function that is captured)."""

    def __init__(self, on=True):
        self.on = int(bool(on))

    def __enter__(self):
        self.old = _lib.lib().dhaug_set_nan_propagation(self.on)
        return self

    def __exit__(self, *exc):
        _lib.lib().dhaug_set_nan_propagation(self.old)


def tn_group_ok(M, N1, N2, colsum_rows):
    """shapes dhaug_gemm_tn_group_bf16 takes (and where it pays: a long batch)"""
    return M >= 1024 and M % 32 == 0 and colsum_rows % 32 == 0 and 1 <= N1 <= 256 and 1 <= N2 <= 256


TN_WIDE_MIN_BLOCKS = int(os.environ.get("DHAUG_TN_WIDE_MIN_BLOCKS", "128"))


def gemm_tn_group(items, max_workgroups=0, phase=0, workspace=None):
    """items: [(A, B, N1, N2, out, colsum | None, colsum_rows, accumulate, M | None, lda | None, ldb | None)] -- the weight
    gradients C_i (+)= A_i^T B_i of several layers in one launch (+ one that sums the partial results).
    max_workgroups: leave CUs to kernels running beside this launch (0: one workgroup per CU).
    phase 1: the contractions only (partial results into `workspace`); phase 2: only their sums into the outputs (same items,
    same workspace; one chunk of at most TN_GROUP_MAX items); 0: both."""
    # layers wider than 256: whole ("wide": one workgroup per 256 x 256 block, which adds into the gradient slot itself) where the
    # chunk they travel in has blocks enough to fill the card without splitting any over the batch; as 256 x 256 blocks, each an
    # item of its own, otherwise (a long batch of few wide layers: the blocks are split over the batch and summed)
    nblk = lambda it: ((it[2] + 255) // 256) * ((it[3] + 255) // 256)
    # items may carry two more fields, (planes_a, planes_b): A / B as the three planes of a split operand (dhaug_tn_layer.planes_a / _b)
    items = [tuple(it) if len(it) == 13 else tuple(it) + (0, 0) for it in items]
    if any(nblk(it) > 1 for it in items):
        flat = []
        for i0 in range(0, len(items), _lib.TN_GROUP_MAX):
            chunk = items[i0:i0 + _lib.TN_GROUP_MAX]
            if phase == 0 and sum(nblk(it) for it in chunk) >= TN_WIDE_MIN_BLOCKS:
                flat.extend(chunk)
                continue
            for (A, B, N1, N2, out, cs, cr, accumulate, M, la, lb, pa, pb) in chunk:
                assert (pa == 0 and pb == 0) or (N1 <= 256 and N2 <= 256), "a wide layer's operands as planes: not built"
                for n0 in range(0, N1, 256):
                    for k0 in range(0, N2, 256):
                        c = cs[n0:] if (cs is not None and k0 == 0) else None
                        flat.append((A[:, n0:], B[:, k0:], min(256, N1 - n0), min(256, N2 - k0), out[n0:, k0:], c,
                                     cr if c is not None else 0, accumulate, A.shape[0] if M is None else M, la, lb, pa, pb))
        items = flat
    assert phase == 0 or (len(items) <= _lib.TN_GROUP_MAX and workspace is not None)
    # the items of ONE launch are summed into their outputs concurrently (a block that is left with one workgroup adds its
    # result into C / colsum with a plain read-modify-write): two items with the same output must not share a launch
    # (include/dhaug.h, dhaug_gemm_tn_group_bf16).  A later contribution to an output waits for a launch of its own.
    seen, later = set(), []
    first = []
    for it in items:
        keys = {("C", _p(it[4]))} | ({("s", _p(it[5]))} if it[5] is not None else set())
        if keys & seen:
            later.append(it)
        else:
            seen |= keys
            first.append(it)
    if later:
        if phase != 0:
            raise RuntimeError("gemm_tn_group: two items of a phased launch share an output")
        gemm_tn_group(first, max_workgroups, 0, workspace)
        gemm_tn_group(later, max_workgroups, 0, workspace)
        return
    for i0 in range(0, len(items), _lib.TN_GROUP_MAX):
        chunk = items[i0:i0 + _lib.TN_GROUP_MAX]
        arr = (_lib.TnLayer * len(chunk))()
        for d, (A, B, N1, N2, out, cs, cr, accumulate, M, la, lb, pa, pb) in zip(arr, chunk):
            d.planes_a, d.planes_b = int(pa), int(pb)
            assert A.dtype == BF16 and B.dtype == BF16 and out.dtype == torch.float32
            d.A, d.lda, d.B, d.ldb = _p(A), A.stride(0) if la is None else la, _p(B), B.stride(0) if lb is None else lb
            d.C, d.ldc, d.colsum_a, d.colsum_rows = _p(out), out.stride(0), _p(cs), cr
            d.M, d.N1, d.N2, d.accumulate = A.shape[0] if M is None else M, N1, N2, int(bool(accumulate))
            d.max_workgroups = int(max_workgroups)
        ws = workspace if workspace is not None else _tn_group_workspace(chunk[0][0].device)
        _lib.call("dhaug_gemm_tn_group_bf16_phase", arr, len(chunk), _p(ws), int(phase), _stream())


def gemm_tn(A, B, N1, N2, out=None, accumulate=False, M=None, lda=None, ldb=None, colsum=None, colsum_rows=None):
    """C[N1,N2] (+)= A[M,N1]^T B[M,N2], bf16 operands, fp32 result; colsum (fp32 [N1], optional) (+)= column sums of A
    (over rows [0, colsum_rows) only when given: a multiple of 128)."""
    assert A.dtype == BF16 and B.dtype == BF16
    M = A.shape[0] if M is None else M
    if out is None:
        out = torch.empty((N1, N2), dtype=torch.float32, device=A.device)
        accumulate = False
    la, lb = A.stride(0) if lda is None else lda, B.stride(0) if ldb is None else ldb
    cr = M if colsum_rows is None else colsum_rows
    if TN256 and N1 == 256 and N2 == 256 and tn_group_ok(M, N1, N2, cr):
        # a long 256 x 256 contraction alone: a group of one (whole-output tiles, dhaug_tn256.hip)
        gemm_tn_group([(A, B, N1, N2, out, colsum, cr, accumulate, M, la, lb)])
        return out
    _lib.call("dhaug_gemm_tn_bf16_rows", _p(A), A.stride(0) if lda is None else lda, _p(B), B.stride(0) if ldb is None else ldb,
              _p(out), out.stride(0), _p(colsum), M if colsum_rows is None else colsum_rows, M, N1, N2, int(accumulate), _stream())
    return out


def colsum(src, N=None, out=None, accumulate=False):
    N = src.shape[1] if N is None else N
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=src.device)
        accumulate = False
    name = "dhaug_colsum_bf16" if src.dtype == BF16 else "dhaug_colsum_f32"
    _lib.call(name, _p(src), src.stride(0), _p(out), src.shape[0], N, int(accumulate), _stream())
    return out


def act_backward(g, y, act, slope=0.0, out=None):
    """g * act'(y), same dtype/shape as g (bf16: widths multiple of 8, row-strided views allowed; fp32: contiguous).
    out: write there (may be g itself)."""
    assert g.dtype == y.dtype and g.shape == y.shape
    if out is None:
        out = torch.empty_like(g) if g.is_contiguous() else torch.empty(g.shape, dtype=g.dtype, device=g.device)
    if g.dtype == BF16:
        _lib.call("dhaug_act_backward_bf16", _p(g), g.stride(0), _p(y), y.stride(0), _p(out), out.stride(0), g.shape[0],
                  g.shape[1], act, float(slope), _stream())
    else:
        assert g.is_contiguous() and y.is_contiguous() and out.is_contiguous()
        _lib.call("dhaug_act_backward_f32", _p(g), _p(y), _p(out), g.numel(), act, float(slope), _stream())
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-4, betas=(0.5, 0.9), eps=1e-8, grad_scale=1.0):
    n = param.numel()
    assert param.is_contiguous() and grad.is_contiguous() and grad.numel() == n
    _lib.call("dhaug_adam_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, lr, betas[0], betas[1], eps,
              int(step), float(grad_scale), _stream())


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, step_dev, lr=1e-4, betas=(0.5, 0.9), eps=1e-8, grad_scale=1.0):
    """Adam step whose count lives on the device (int32 tensor, advanced here): replayable inside a captured graph"""
    n = param.numel()
    assert param.is_contiguous() and grad.is_contiguous() and grad.numel() == n and step_dev.dtype == torch.int32
    _lib.call("dhaug_counter_add", _p(step_dev), 1, _stream())
    _lib.call("dhaug_adam_step_dev", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, lr, betas[0], betas[1], eps,
              _p(step_dev), float(grad_scale), _stream())


# --------------------------------------------------------------------------------------- WGAN-GP arithmetic
def gp_assemble(real, fake, alpha, out=None, bf16_rows=False):
    """(3B, W) fp32 = [real; fake; alpha * real + (1 - alpha) * fake].  bf16_rows (W a multiple of 16): the real / fake rows also as
    bf16 (2B, W), attached as out._dhaug_bf16_rows -- what cast_pad_bf16(out[:2B], W) would make, from the same launch"""
    r = _dev(real, torch.float32, "gp_assemble")
    r = r.reshape(r.shape[0], -1)
    f = _dev(fake, torch.float32, "gp_assemble").reshape(r.shape)
    a = _dev(alpha, torch.float32, "gp_assemble").reshape(-1)
    B, W = r.shape
    assert a.shape[0] == B
    if out is None:
        out = torch.empty((3 * B, W), dtype=torch.float32, device=r.device)
    if bf16_rows and W % 16 == 0:
        xb = torch.empty((2 * B, W), dtype=BF16, device=r.device)
        _lib.call("dhaug_gp_assemble_bf16", _p(r), _p(f), _p(a), _p(out), _p(xb), W, B, W, _stream())
        out._dhaug_bf16_rows = xb
        return out
    _lib.call("dhaug_gp_assemble", _p(r), _p(f), _p(a), _p(out), B, W, _stream())
    return out


def gp_penalty(grad, coef, bf16=False):
    """per-row (||g|| - 1)^2 and the penalty's cotangent coef * (n - 1) / n * g.  bf16 (W a multiple of 16): v also as bf16,
    attached as v._dhaug_bf16 (= cast_pad_bf16(v, W), from the same launch)"""
    g = _dev(grad, torch.float32, "gp_penalty")
    B, W = g.shape
    v = torch.empty_like(g)
    pen = torch.empty((B,), dtype=torch.float32, device=g.device)
    if bf16 and W % 16 == 0:
        vb = torch.empty((B, W), dtype=BF16, device=g.device)
        _lib.call("dhaug_gp_penalty_bf16", _p(g), _p(v), _p(vb), W, _p(pen), B, W, float(coef), _stream())
        v._dhaug_bf16 = vb
        return v, pen
    _lib.call("dhaug_gp_penalty", _p(g), _p(v), _p(pen), B, W, float(coef), _stream())
    return v, pen


def d3_penalty(pose16, grad_kcs, grad_pose, coef):
    """The 3D critic's penalty step in one launch (dhaug_d3_penalty): from the x_hat poses (N, 48) fp32 and the two branches' input
    cotangents (N, 30) / (N, 48) fp32 -> (tangent input of the KCS branch (N, 32) bf16, of the pose branch (N, 48) bf16, pen (N) fp32)
    = kcs_backward + add_f32 + gp_penalty + kcs_jvp + the two bf16 casts (same operations, same order)."""
    x = _dev(pose16, torch.float32, "d3_penalty").reshape(-1, 48)
    N = x.shape[0]
    gk = _dev(grad_kcs, torch.float32, "d3_penalty").reshape(N, 30)
    gp = _dev(grad_pose, torch.float32, "d3_penalty").reshape(N, 48)
    tk = torch.empty((N, 32), dtype=BF16, device=x.device)
    tv = torch.empty((N, 48), dtype=BF16, device=x.device)
    pen = torch.empty((N,), dtype=torch.float32, device=x.device)
    _lib.call("dhaug_d3_penalty", _p(x), _p(gk), _p(gp), float(coef), _p(tk), _p(tv), _p(pen), N, _stream())
    return tk, tv, pen


CRITIC_SCALARS_SCRATCH = 192      # floats: DHAUG_CRITIC_SCALARS_SCRATCH of include/dhaug.h (tests/test_cpu_boundary.py compares)


def critic_scalars(logits, pen, B, lam):
    """(5,) fp32: D_real, D_fake, GP, Wasserstein_D, D_cost (logit means over B rows per half, penalty mean over len(pen))"""
    # result (5 floats, padded to 8 so that the scratch stays 32-byte aligned) + the partial sums of stage 1
    buf = torch.empty((8 + CRITIC_SCALARS_SCRATCH,), dtype=torch.float32, device=logits.device)
    scratch = buf[8:]
    assert scratch.numel() >= CRITIC_SCALARS_SCRATCH
    _lib.call("dhaug_critic_scalars", _p(logits), logits.stride(0), _p(pen), B, pen.numel(), float(lam), _p(buf), _p(scratch), _stream())
    return buf[:5]


def rank1_mask(seed, w_col, mask, n, act, slope=0.0, out=None):
    """bf16 (M, pad): bf16(seed[r] * w[c]) * act'(mask[r][c]) -- the first backward step through a 1-wide logit layer.
    seed (M, >=1) bf16 (column 0), w_col: bf16 view with the layer's n weights along dim 0 (any stride)"""
    assert seed.dtype == BF16 and w_col.dtype == BF16 and mask.dtype == BF16
    M, pad = mask.shape[0], min(mask.shape[1], ceil_to(n, 16))
    if out is None:
        out = torch.empty((M, ceil_to(n, 16)), dtype=BF16, device=mask.device)
    bits = getattr(mask, "_dhaug_bits", None)
    if bits is not None and DBITS and n == 256 and act != 0 and out.stride(0) % 8 == 0 and out.shape[1] >= 256:
        assert bits.device == out.device, "sign bits must live on the operands' device"
        _lib.call("dhaug_rank1_bits_bf16", _p(seed), seed.stride(0), _p(w_col), w_col.stride(0), _p(bits), _p(out), out.stride(0), M,
                  act, float(slope), _stream())
        return out
    _lib.call("dhaug_rank1_mask_bf16", _p(seed), seed.stride(0), _p(w_col), w_col.stride(0), _p(mask), mask.stride(0), _p(out),
              out.stride(0), M, n, pad, act, float(slope), _stream())
    return out


def top_backward_ok(M, n0, nc, masks, bits_cols):
    """dhaug_critic_top_backward_bf16 covers: whole 64-row tiles, a merge block of at most 112 features behind a concatenation of two
    256-wide branches whose masks are sign-bit arrays, bf16 activations with 16-byte aligned rows"""
    return (TOP_FUSED and M >= 64 and M % 64 == 0 and 1 <= n0 <= 112 and nc == 512 and bits_cols is not None and len(bits_cols) == 2
            and all(b is not None for b in bits_cols)
            and all(t.dtype == BF16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.shape[1] >= 112
                    and t.data_ptr() % 16 == 0 for t in masks))


def critic_top_backward(seed, w_out_col, m1, mh, m0, W2nn, W1nn, Wmnn, bits_cols, n0, act, slope=0.0, gcat=None):
    """sweep 2 through the top of a branch critic in ONE launch (dhaug_critic_top_backward_bf16): returns (gz_m2, gz_m1, gz_m0, gcat) --
    bf16 (M, 112) cotangents at the logit layer's input / the merge block's fc1 pre-activation / the merge layer's pre-activation, and
    (M, 512) at the branches' outputs.  seed (M, >= 1) bf16; w_out_col: the logit layer's weights along dim 0; m1, mh, m0: saved
    activations; W2nn, W1nn (n0, >= 112), Wmnn (512, >= 112): operand copies whose rows are the products' outputs."""
    M = m1.shape[0]
    dev = m1.device
    g2, g1, g0 = (torch.empty((M, 112), dtype=BF16, device=dev) for _ in range(3))
    if gcat is None:
        gcat = torch.empty((M, 512), dtype=BF16, device=dev)
    assert gcat.dtype == BF16 and gcat.stride(1) == 1 and gcat.shape[0] == M and gcat.shape[1] >= 512
    assert all(b.device == dev for b in bits_cols), "sign bits must live on the operands' device"
    d = _lib.TopDesc()
    d.seed, d.ld_seed, d.wout, d.ld_wout = _p(seed), seed.stride(0), _p(w_out_col), w_out_col.stride(0)
    d.x, d.ldx = None, 0
    d.m1, d.mh, d.m0, d.ld_m = _p(m1), _p(mh), _p(m0), m1.stride(0)
    assert mh.stride(0) == m1.stride(0) and m0.stride(0) == m1.stride(0)
    d.w2, d.ldw2, d.w1, d.ldw1, d.wm, d.ldwm = _p(W2nn), W2nn.stride(0), _p(W1nn), W1nn.stride(0), _p(Wmnn), Wmnn.stride(0)
    d.bits0, d.bits1 = _p(bits_cols[0]), _p(bits_cols[1])
    d.g2, d.g1, d.g0, d.ld_g, d.gcat, d.ld_gcat = _p(g2), _p(g1), _p(g0), 112, _p(gcat), gcat.stride(0)
    d.M, d.n0, d.nc, d.mask_act, d.mask_slope = M, n0, 512, act, float(slope)
    _lib.call("dhaug_critic_top_backward_bf16", ctypes.byref(d), _stream())
    return g2, g1, g0, gcat


def critic_top_tangent(ucat, m0, mh, m1, Wm_nt, W1_nt, W2_nt, n0, act, slope=0.0):
    """sweep 3 through the top of a branch critic in ONE launch (dhaug_critic_top_tangent_bf16), IN PLACE: the rows of the saved
    activations m0, mh, m1 (the interpolated rows' views) are overwritten with the tangents um0, umh, um1, which are returned.
    ucat (M, 512) bf16; Wm_nt (n0, >= 512), W1_nt / W2_nt (n0, >= 112): the layers' "nt" operand copies."""
    M = ucat.shape[0]
    assert ucat.dtype == BF16 and ucat.stride(1) == 1 and ucat.shape[1] >= 512
    assert mh.stride(0) == m0.stride(0) and m1.stride(0) == m0.stride(0)
    d = _lib.TopDesc()
    d.seed, d.ld_seed, d.wout, d.ld_wout = None, 0, None, 0
    d.x, d.ldx = _p(ucat), ucat.stride(0)
    d.m1, d.mh, d.m0, d.ld_m = _p(m1), _p(mh), _p(m0), m0.stride(0)
    d.w2, d.ldw2, d.w1, d.ldw1, d.wm, d.ldwm = _p(W2_nt), W2_nt.stride(0), _p(W1_nt), W1_nt.stride(0), _p(Wm_nt), Wm_nt.stride(0)
    d.bits0, d.bits1, d.g2, d.g1, d.g0, d.ld_g, d.gcat, d.ld_gcat = None, None, None, None, None, 0, None, 0
    d.M, d.n0, d.nc, d.mask_act, d.mask_slope = M, n0, 512, act, float(slope)
    _lib.call("dhaug_critic_top_tangent_bf16", ctypes.byref(d), _stream())
    return m0, mh, m1


def top_tangent_ok(M, n0, nc, ucat, masks):
    return (TOP_FUSED and M >= 64 and M % 64 == 0 and 1 <= n0 <= 112 and nc == 512 and ucat.dtype == BF16 and ucat.dim() == 2
            and ucat.stride(1) == 1 and ucat.stride(0) % 8 == 0 and ucat.shape[1] >= 512 and ucat.data_ptr() % 16 == 0
            and all(t.dtype == BF16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.shape[1] >= 112
                    and t.data_ptr() % 16 == 0 and t.stride(0) == masks[0].stride(0) for t in masks))


def add_f32(a, b):
    """a + b (fp32, same shape)"""
    a, b = _dev(a, torch.float32, "add_f32"), _dev(b, torch.float32, "add_f32")
    assert a.shape == b.shape
    out = torch.empty_like(a)
    _lib.call("dhaug_add_f32", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def frame_diff(x, R, in_w, w=None, adjoint=False):
    """clips (rows, R*in_w) -> adjacent-frame differences (rows, (R-1)*w) over the first w columns of every frame; adjoint:
    (rows, (R-1)*w) -> (rows, R*in_w), the transposed map"""
    w = in_w if w is None else w
    v = _dev(x, torch.float32, "frame_diff")
    rows = v.shape[0]
    out = torch.empty((rows, R * in_w if adjoint else (R - 1) * w), dtype=torch.float32, device=v.device)
    _lib.call("dhaug_frame_diff", _p(v), _p(out), rows, R, in_w, w, int(bool(adjoint)), _stream())
    return out


def frame_reverse(x, R, w):
    """clips (rows, R*w) -> the frames of every clip in reverse order (its own transpose: also the backward map)"""
    v = _dev(x, torch.float32, "frame_reverse")
    rows = v.shape[0]
    assert v.shape[1] == R * w
    out = torch.empty_like(v)
    _lib.call("dhaug_frame_reverse", _p(v), _p(out), rows, R, w, _stream())
    return out


def weighted_means(arrays, weights):
    """0-dim fp32 tensor sum_i weights[i] * mean(arrays[i]) (fp32 device tensors, contiguous), one launch"""
    n = len(arrays)
    assert 1 <= n <= 16 and len(weights) == n
    arrs = [_dev(a, torch.float32, "weighted_means").reshape(-1) for a in arrays]
    out = torch.empty((1,), dtype=torch.float32, device=arrs[0].device)
    P = (_vp * n)(*[a.data_ptr() for a in arrs])
    C = (ctypes.c_int64 * n)(*[a.numel() for a in arrs])
    W = (ctypes.c_float * n)(*[float(w) for w in weights])
    _lib.call("dhaug_weighted_means", P, C, W, n, _p(out), _stream())
    return out.reshape(())
