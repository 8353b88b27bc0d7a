"""Hot-path camera functions of R/common/camera.py, backed by the HIP kernels of csrc/dhaug_pose.hip."""
import numpy as np
import torch

from .. import autograd_ops as A
from .. import ops


def normalize_screen_coordinates(point, w, h):
    """R/common/camera.py:11-16 (host-side, on the principal point only)."""
    point = np.array(point, dtype=np.float64, copy=True)
    point[..., 0] = point[..., 0] / w * 2 - 1
    point[..., 1] = point[..., 1] / w * 2 - h / w
    return point


def camera_params9(intrinsic):
    """f(2) c(2) k(3) p(2) in normalised screen units, as assembled at
    R/models_Fk_GAN/model_fk_gan_train.py:352-364."""
    res_w, res_h = float(intrinsic["res_w"]), float(intrinsic["res_h"])
    f = np.array(intrinsic["focal_length"]) / res_w * 2.0
    c = normalize_screen_coordinates(np.array(intrinsic["center"]), w=res_w, h=res_h).astype("float32")
    return [float(v) for v in (*f, *c, *intrinsic["radial_distortion"], *intrinsic["tangential_distortion"])]


def GAN_torch_world_to_camera(X, R, t):
    """qinverse + qrot of X - t with one shared camera (R (1,4), t (1,3)); R/common/camera.py:36-38."""
    c3, _ = A.W2CProjectFn.apply(X.reshape(-1, 16, 3), tuple(float(v) for v in R.reshape(-1)[:4].tolist()),
                                 tuple(float(v) for v in t.reshape(-1)[:3].tolist()), (1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
    return c3


def world_to_camera_project(X, quat, trans, cam9):
    """fused GAN_torch_world_to_camera + project_to_2d -> (cam3d, proj2d); differentiable w.r.t. X."""
    return A.W2CProjectFn.apply(X.reshape(-1, 16, 3), tuple(quat), tuple(trans), tuple(cam9))


def GAN_torch_camera_to_world_batch(X, R, t):
    """per-sample quaternion / translation; R/common/camera.py:53-59."""
    return ops.camera_to_world(X, R, t)
