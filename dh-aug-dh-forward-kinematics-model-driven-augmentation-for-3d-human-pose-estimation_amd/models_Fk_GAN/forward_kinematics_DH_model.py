"""Drop-in for R/models_Fk_GAN/forward_kinematics_DH_model.py (class Forward_Kinematics_DH_Model).

The 33 dh_matrix builds, 46 bmm, global rotation, 51 column scatters and the root add of the reference's
change_3d_joint_angle (:562-822) are ONE HIP kernel (dhaug_fk_forward).  Unlike the reference the object holds
no per-batch constant tensors, so any number of poses can be passed (the reference is fixed to
args.batch_size * frames at construction, :274-321)."""
import numpy as np
import torch

from .. import autograd_ops as A
from .. import ops
from .video_mode_operate import frames_from_args

used_16key_15bone_len_table = [(5, 6), (2, 3), (4, 5), (1, 2), (0, 4), (0, 1), (0, 7), (7, 8), (8, 10), (8, 13),
                               (10, 11), (13, 14), (11, 12), (14, 15), (8, 9)]
H36M_32_To_16_Table = [0, 1, 2, 3, 6, 7, 8, 12, 13, 15, 17, 18, 19, 25, 26, 27]
H36M_POINTS_LEFT = [6, 7, 8, 17, 18, 19]
H36M_POINTS_RIGHT = [1, 2, 3, 25, 26, 27]

_BONE_KW = ["left_small_leg_len", "right_small_leg_len", "left_big_leg_len", "right_big_leg_len", "left_hip_len",
            "right_hip_len", "waist_len", "thorax_len", "left_shoulder_len", "right_shoulder_len", "left_big_arm_len",
            "right_big_arm_len", "left_small_arm_len", "right_small_arm_len", "neck_len"]


def pack_angles(right_leg, left_leg, body, right_hand, left_hand, global_rot):
    """reference kwargs -> (N,37) generator_angle layout (R/models_Fk_GAN/Fk_generator.py:179-186)."""
    N = right_leg.shape[0]
    zero = torch.zeros((N, 1), dtype=torch.float32, device=right_leg.device)
    return torch.cat([right_leg, left_leg, body, right_hand, left_hand, zero, global_rot], dim=1)


class Forward_Kinematics_DH_Model():
    def __init__(self, args, train_subjects, dataset):
        self.args = args
        self.train_subjects = train_subjects
        self.GAN_BATCH_SIZE = args.batch_size
        self.dataset = dataset
        self.random = np.random.RandomState(args.random_seed)       # :204-205
        self.real_used_num = frames_from_args(args)
        self.device = torch.device("cuda") if torch.cuda.is_available() else None

    def random_state(self):
        return self.random

    def set_random_state(self, random):
        self.random = random

    def change_3d_joint_angle(self, left_leg_joints_angle, right_leg_joints_angle, body_joints_angle,
                              left_hand_joints_angle, right_hand_joints_angle, generator_global_rot_3d_pos_angle,
                              left_small_leg_len, right_small_leg_len, left_big_leg_len, right_big_leg_len,
                              left_hip_len, right_hip_len, waist_len, thorax_len, left_shoulder_len,
                              right_shoulder_len, left_big_arm_len, right_big_arm_len, left_small_arm_len,
                              right_small_arm_len, neck_len, root_3d_pos):
        lens = [left_small_leg_len, right_small_leg_len, left_big_leg_len, right_big_leg_len, left_hip_len,
                right_hip_len, waist_len, thorax_len, left_shoulder_len, right_shoulder_len, left_big_arm_len,
                right_big_arm_len, left_small_arm_len, right_small_arm_len, neck_len]
        if not torch.is_tensor(left_leg_joints_angle):
            # scalar / numpy branch of the reference (:366-560): one pose, returns ndarray (32,3) float32
            if self.device is None:
                raise RuntimeError("Forward_Kinematics_DH_Model needs a GPU (no CPU fallback exists)")
            f = lambda v, n: torch.as_tensor(np.asarray(v, dtype=np.float32).reshape(1, n), device=self.device)
            ang = pack_angles(f(right_leg_joints_angle, 5), f(left_leg_joints_angle, 5), f(body_joints_angle, 13),
                              f(right_hand_joints_angle, 5), f(left_hand_joints_angle, 5),
                              f(generator_global_rot_3d_pos_angle, 3))
            bl = f([float(v) for v in lens], 15)
            return ops.fk_forward(ang, bl, f(root_3d_pos, 3), 32)[0].cpu().numpy().astype(np.float32)

        ang = pack_angles(right_leg_joints_angle, left_leg_joints_angle, body_joints_angle, right_hand_joints_angle,
                          left_hand_joints_angle, generator_global_rot_3d_pos_angle)
        bl = torch.stack([v.reshape(-1) for v in lens], dim=1)
        root = root_3d_pos.reshape(-1, 3)
        if not (ang.requires_grad or bl.requires_grad or root.requires_grad):
            return ops.fk_forward(ang, bl, root, 32)
        out16 = A.FkFn.apply(ang.contiguous(), bl.contiguous(), root.contiguous())
        out32 = root.reshape(-1, 1, 3).expand(-1, 32, -1).clone()           # rows the reference never writes = root
        idx = torch.as_tensor(H36M_32_To_16_Table, device=out16.device)
        out32 = out32.index_copy(1, idx, out16)
        out32 = out32.index_copy(1, torch.as_tensor([14], device=out16.device), out16[:, 9:10])   # slot 14 = slot 15
        return out32

    def fk16(self, angles37, bone_len, root):
        """(N,37),(N,15),(N,3) -> (N,16,3): the fused form the generator uses (FK + 32->16 gather)."""
        return A.FkFn.apply(angles37.contiguous(), bone_len.contiguous(), root.contiguous())

    def init_Fk_DH_angle(self):
        """T-pose with the default lengths (:824-858)."""
        z5, z13 = [0.0] * 5, [0.0] * 13
        return self.change_3d_joint_angle(
            left_leg_joints_angle=z5, right_leg_joints_angle=z5, body_joints_angle=z13, left_hand_joints_angle=z5,
            right_hand_joints_angle=z5, generator_global_rot_3d_pos_angle=(0.0, 0.0, 0.0), left_small_leg_len=0.5,
            right_small_leg_len=0.5, left_big_leg_len=0.6, right_big_leg_len=0.6, left_hip_len=0.25, right_hip_len=0.25,
            waist_len=0.25, thorax_len=0.2, left_shoulder_len=0.4, right_shoulder_len=0.4, left_big_arm_len=0.4,
            right_big_arm_len=0.4, left_small_arm_len=0.35, right_small_arm_len=0.35, neck_len=0.15,
            root_3d_pos=(0.0, 0.0, 0.0))
