"""ctypes binding of libdhaug.so (include/dhaug.h).  Loaded lazily; never falls back to anything else."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DHAUG_LIB", os.path.join(_HERE, "lib", "libdhaug.so"))

_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
_u64 = ctypes.c_uint64

# name -> argtypes, in the order of include/dhaug.h
SIGNATURES = {
    "dhaug_fk_forward": [_vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "dhaug_fk_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "dhaug_gen_tail_forward": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "dhaug_gen_tail_forward_critics": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _u64, _u64, _vp, _i64, _i32, _i32, _vp],
    "dhaug_gen_tail_backward": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "dhaug_bone_length": [_vp, _vp, _i64, _vp],
    "dhaug_kcs_forward": [_vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "dhaug_center_kcs_forward": [_vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "dhaug_kcs_backward": [_vp, _vp, _vp, _i64, _i32, _vp],
    "dhaug_kcs_jvp": [_vp, _vp, _vp, _i64, _i32, _vp],
    "dhaug_world_to_camera_project": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "dhaug_world_to_camera_project_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "dhaug_camera_to_world": [_vp, _vp, _vp, _vp, _i64, _vp],
    "dhaug_bone_length_swap": [_vp, _vp, _vp, _i64, _vp],
    "dhaug_project_to_2d": [_vp, _vp, _vp, _i64, _vp],
    "dhaug_center_flip": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "dhaug_center_flip_backward": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "dhaug_gemm_bf16_dmask": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _f32, _vp, _i64, _i64, _i64, _i64, _vp],
    "dhaug_gemm_bf16_dmask_pad": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _f32, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
    "dhaug_gemm_bf16_dbits": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _f32, _vp, _i64, _i64, _vp],
    "dhaug_gemm_block2_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _i64, _vp, _i64, _i64, _vp],
    "dhaug_set_workgroup_cap": [_i32],
    "dhaug_set_nan_propagation": [_i32],
    "dhaug_rank1_bits_bf16": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _f32, _vp],
    "dhaug_gemm_bf16_dbits_wide": [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _i64, _i64, _i64, _i64, _vp],
    "dhaug_gemm_bf16": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i32,
                        _f32, _vp],
    "dhaug_gemm_tn_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp],
    "dhaug_gemm_tn_bf16_rows": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "dhaug_cast_pad_bf16": [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp],
    "dhaug_cast_transpose_bf16": [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp],
    "dhaug_split_bf16": [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp],
    "dhaug_split_f16": [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp],
    "dhaug_gemm_bf16x6_planes": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _f32, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _f32, _vp],
    "dhaug_gemm_f16x3": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _f32, _vp],
    "dhaug_gemm_f16x3_planes": [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _f32, _vp],
    "dhaug_colsum_f32": [_vp, _i64, _vp, _i64, _i64, _i32, _vp],
    "dhaug_colsum_bf16": [_vp, _i64, _vp, _i64, _i64, _i32, _vp],
    "dhaug_act_backward_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i32, _f32, _vp],
    "dhaug_act_backward_f32": [_vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "dhaug_adam_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _vp],
    "dhaug_adam_step_dev": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _vp, _f32, _vp],
    "dhaug_counter_add": [_vp, _i32, _vp],
    "dhaug_frame_diff": [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "dhaug_repack_weights": [_vp, _i32, _vp],
    "dhaug_gemm_bf16_group": [_vp, _i32, _vp],
    "dhaug_gemm_bf16_dmask_f32": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _f32, _vp, _i64, _i64, _i64, _i64, _vp],
    "dhaug_adam_repack_step": [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _f32, _vp, _i32, _i64, _vp, _i32, _vp],
    "dhaug_gp_assemble": [_vp, _vp, _vp, _vp, _i64, _i64, _vp],
    "dhaug_gp_penalty": [_vp, _vp, _vp, _i64, _i64, _f32, _vp],
    "dhaug_gp_assemble_bf16": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "dhaug_gp_penalty_bf16": [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _f32, _vp],
    "dhaug_d3_penalty": [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _vp],
    "dhaug_critic_scalars": [_vp, _i64, _vp, _i64, _i64, _f32, _vp, _vp, _vp],
    "dhaug_rank1_mask_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _f32, _vp],
    "dhaug_add_f32": [_vp, _vp, _vp, _i64, _vp],
    "dhaug_frame_reverse": [_vp, _vp, _i64, _i32, _i32, _vp],
    "dhaug_weighted_means": [ctypes.POINTER(_vp), ctypes.POINTER(_i64), ctypes.POINTER(_f32), _i32, _vp, _vp],
}

class MlpUnit(ctypes.Structure):
    """struct dhaug_mlp_unit (include/dhaug.h)"""
    _fields_ = [("kind", _i32), ("flags", _i32), ("src", _i32), ("dst", _i32), ("res", _i32), ("src2", _i32),
                ("ksteps2", _i32), ("ksteps", _i32), ("n", _i32), ("act", _i32), ("slope", _f32), ("cols", _i32),
                ("ld", _i64), ("g", _vp), ("w", _vp), ("w2", _vp), ("bias", _vp), ("save", _vp), ("save_ld", _i64), ("bits", _vp),
                ("save_rows", _i64)]


class WfragDesc(ctypes.Structure):
    """struct dhaug_wfrag_desc (include/dhaug.h)"""
    _fields_ = [("W", _vp), ("ldw", _i64), ("dst", _vp), ("bias", _vp), ("bias_dst", _vp), ("dot_dst", _vp), ("N", _i32),
                ("K", _i32), ("k0", _i32), ("ksteps", _i32)]


class RepackDesc(ctypes.Structure):
    """struct dhaug_repack_desc (include/dhaug.h)"""
    _fields_ = [("W", _vp), ("nt", _vp), ("nn", _vp), ("N", _i32), ("K", _i32), ("Kp", _i32), ("Np", _i32)]


class GemmDesc(ctypes.Structure):
    """struct dhaug_gemm_desc (include/dhaug.h)"""
    _fields_ = [("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("bias", _vp), ("residual", _vp), ("ld_res", _i64),
                ("residual_f32", _vp), ("ld_res_f32", _i64), ("c_bf16", _vp), ("ldc_bf16", _i64), ("n_pad_zero", _i64),
                ("c_f32", _vp), ("ldc_f32", _i64), ("M", _i64), ("N", _i64), ("K", _i64), ("act", _i32), ("slope", _f32),
                ("dmask", _vp), ("ld_dmask", _i64), ("dmask_act", _i32), ("dmask_slope", _f32)]


class AdamDesc(ctypes.Structure):
    """struct dhaug_adam_desc (include/dhaug.h)"""
    _fields_ = [("off", _i64), ("len", _i64), ("nt", _vp), ("N", _i32), ("K", _i32), ("Kp", _i32), ("pad_", _i32), ("item0", _i64)]


class TnLayer(ctypes.Structure):
    """struct dhaug_tn_layer (include/dhaug.h)"""
    _fields_ = [("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64), ("colsum_a", _vp),
                ("colsum_rows", _i64), ("M", _i64), ("N1", _i32), ("N2", _i32), ("accumulate", _i32), ("max_workgroups", _i32),
                ("planes_a", _i32), ("planes_b", _i32)]


class Block2(ctypes.Structure):
    """struct dhaug_block2 (include/dhaug.h)"""
    _fields_ = [("W1", _vp), ("ldw1", _i64), ("W2", _vp), ("ldw2", _i64), ("bits1", _vp), ("bits2", _vp), ("Y1", _vp), ("ldy1", _i64),
                ("Y2", _vp), ("ldy2", _i64)]


class TopDesc(ctypes.Structure):
    """struct dhaug_top_desc (include/dhaug.h)"""
    _fields_ = [("seed", _vp), ("ld_seed", _i64), ("wout", _vp), ("ld_wout", _i64), ("x", _vp), ("ldx", _i64), ("m1", _vp), ("mh", _vp),
                ("m0", _vp), ("ld_m", _i64), ("w2", _vp), ("ldw2", _i64), ("w1", _vp), ("ldw1", _i64), ("wm", _vp), ("ldwm", _i64),
                ("bits0", _vp), ("bits1", _vp), ("g2", _vp), ("g1", _vp), ("g0", _vp), ("ld_g", _i64), ("gcat", _vp), ("ld_gcat", _i64),
                ("M", _i64), ("n0", _i64), ("nc", _i64), ("mask_act", _i32), ("mask_slope", _f32)]


SIGNATURES["dhaug_critic_top_backward_bf16"] = [ctypes.POINTER(TopDesc), _vp]
SIGNATURES["dhaug_critic_top_tangent_bf16"] = [ctypes.POINTER(TopDesc), _vp]
BLOCK2_MAX = 3
SIGNATURES["dhaug_gemm_block2_stack_bf16"] = [_vp, _i64, ctypes.POINTER(Block2), _i32, _i32, _f32, _i64, _vp]
TN_GROUP_MAX = 42
TN_GROUP_WORKSPACE_FLOATS = 256 * (256 * 256 + 256)
SIGNATURES["dhaug_gemm_tn_group_bf16"] = [ctypes.POINTER(TnLayer), _i32, _vp, _vp]
SIGNATURES["dhaug_gemm_tn_group_bf16_phase"] = [ctypes.POINTER(TnLayer), _i32, _vp, _i32, _vp]
SIGNATURES["dhaug_pack_wfrag"] = [_vp, _i64, _vp, _i64, _i64, _i64, _vp]
SIGNATURES["dhaug_pack_wfrag_batch"] = [_vp, _i32, _vp]
SIGNATURES["dhaug_mlp_forward"] = [ctypes.POINTER(MlpUnit), _i32, _i64, _vp]
SIGNATURES["dhaug_pack_wfrag_f16x2"] = [_vp, _i64, _vp, _i64, _i64, _i64, _vp]
SIGNATURES["dhaug_pack_wfrag_f16x2_t16"] = [_vp, _i64, _vp, _i64, _i64, _i64, _vp]
SIGNATURES["dhaug_mlp_forward_x3"] = [ctypes.POINTER(MlpUnit), _i32, _i64, _vp]

ERRORS = {-1: "DHAUG_EINVAL (bad argument)", -2: "DHAUG_EALIGN (alignment contract violated)",
          -3: "DHAUG_EUNSUPPORTED (shape not implemented)"}

_lib = None


def lib():
    """The loaded library.  Raises RuntimeError if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libdhaug.so not found at %s: build it with `python __graft_entry__.py build` "
                               "(there is no fallback path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.dhaug_version.restype = ctypes.c_int
        L.dhaug_arch.restype = ctypes.c_char_p
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        if os.environ.get("DHAUG_NAN_PROPAGATION", "") in ("1", "true", "yes"):
            L.dhaug_set_nan_propagation(1)         # fused inference programs: NaN / inf reach the logit (include/dhaug.h)
        _lib = L
    return _lib


def check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (name, ERRORS.get(rc, "hipError_t %d" % rc)))


_fn = {}
CALLS = [0]          # C-ABI calls made by this process (bench.py reports calls per step)


def call(name, *args):
    f = _fn.get(name)
    if f is None:
        f = _fn[name] = getattr(lib(), name)
    CALLS[0] += 1
    rc = f(*args)
    if rc != 0:
        check(rc, name)
