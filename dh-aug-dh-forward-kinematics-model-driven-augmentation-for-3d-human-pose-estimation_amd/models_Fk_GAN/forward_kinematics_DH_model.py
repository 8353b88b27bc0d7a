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

    # ------------------------------------------------------------------ 'normal' (non-GAN) augmentation mode
    # R/models_Fk_GAN/forward_kinematics_DH_model.py:867-929: random frame of the training set -> bone lengths / root
    def get_dataSet_3d_and_2d_pose(self, dataset, pix_2d):
        world_3d = {}
        for subject in dataset.subjects():
            world_3d[subject] = {}
            for action in dataset[subject].keys():
                anim = dataset[subject][action]
                world_3d[subject][action] = {cam_idx: anim['positions'] for cam_idx, _ in enumerate(pix_2d[subject][action])}
        self.dataSet_world_3d_pos = world_3d
        self.dataSet_2d_pos = pix_2d

    def my_random_get_sigle_frame_data(self):
        subject = self.train_subjects[self.random.randint(0, len(self.train_subjects))]
        actions = list(self.dataSet_world_3d_pos[subject].keys())
        action = actions[self.random.randint(0, len(actions))]
        cams = list(self.dataSet_world_3d_pos[subject][action].keys())
        cam = cams[self.random.randint(0, len(cams))]
        frame = self.random.randint(0, self.dataSet_world_3d_pos[subject][action][cam].shape[0])
        return subject, action, cam, frame

    def get_bone_len_from_dataSet(self):
        s, a, c, f = self.my_random_get_sigle_frame_data()
        pose = np.asarray(self.dataSet_world_3d_pos[s][a][c][f])
        self.record_bone_len = [float(np.linalg.norm(pose[p] - pose[q])) for p, q in used_16key_15bone_len_table]

    def get_root_3d_pos_from_dataSet(self):
        s, a, c, f = self.my_random_get_sigle_frame_data()
        self.root_3d_pos = np.array(self.dataSet_world_3d_pos[s][a][c][f][0], copy=True)

    # joint limits of the 'normal' sampler (slots 0..33; slot 23 is skipped) and of its global rotation (:933-973)
    _NORMAL_LO = [-90, -90, -45, -135, 0, -45, -45, -45, -135, 0, -25, -10, -20, -20, -10, -25, -20, 0, -20, -90, -20,
                  -45, 0, None, -135, -135, -45, 0, 0, -45, -45, -45, 0, 0]
    _NORMAL_HI = [45, 45, 120, 0, 0, 90, 90, 120, 0, 0, 25, 90, 20, 20, 45, 25, 20, 0, 20, 90, 90, 45, 0, None, 45,
                  45, 180, 135, 0, 135, 135, 180, 135, 0]
    _NORMAL_GLOBAL = [(-20, 20), (-20, 20), (-180, 180)]

    def handler_but_generater(self):
        """'normal' mode sampler (:931-1152): per frame a random subset of DOF is drawn from N((lo+hi)/2, 60) clipped to
        the joint limits (frame 0 = T-pose), bone lengths / root come from random training frames, bone lengths are
        jittered, and every pose goes through FK.  The host-side RNG call sequence of the reference is kept (same
        numpy stream -> same angles); the reference's generator_whole_number sequential numpy FK evaluations are ONE
        launch of the fused FK kernel.  Returns (pos (N,32,3) float32 ndarray, angles (N,33), global_rot list,
        bone_len list, root list) like the reference."""
        if self.device is None:
            raise RuntimeError("Forward_Kinematics_DH_Model needs a GPU (no CPU fallback exists)")
        args, rnd = self.args, self.random
        N = args.generator_whole_number
        self.generator_3d_pos_angle, self.generator_global_rot_3d_pos_angle = [], []
        self.generator_bone_len, self.generator_root = [], []
        if not hasattr(self, "record_bone_len") or self.record_bone_len is None or len(self.record_bone_len) == 0:
            self.record_bone_len = [0.5, 0.5, 0.6, 0.6, 0.25, 0.25, 0.25, 0.2, 0.4, 0.4, 0.4, 0.4, 0.35, 0.35, 0.15]
        if not hasattr(self, "root_3d_pos"):
            self.root_3d_pos = np.array([0, 0, 0])
        for frame in range(N):
            if args.generator_choose_BoneLen:
                self.get_bone_len_from_dataSet()
            self.generator_bone_len.append(self.record_bone_len)
            if args.generator_choose_root_pos:
                self.get_root_3d_pos_from_dataSet()
            self.generator_root.append(self.root_3d_pos)
            k = rnd.randint(0, 34)
            chosen = set(rnd.choice(np.arange(34), size=k, replace=False).tolist())
            ang = []
            for j in range(34):
                if j == 23:
                    continue
                if j in chosen and frame > 0:
                    lo, hi = self._NORMAL_LO[j], self._NORMAL_HI[j]
                    ang.append(min(max(rnd.normal((lo + hi) / 2, 60), lo), hi))
                else:
                    ang.append(0)
            glob = []
            for lo, hi in self._NORMAL_GLOBAL:
                if frame > 0 and args.generator_global_rot:
                    glob.append(min(max(rnd.normal((lo + hi) / 2, 60), lo), hi))
                else:
                    glob.append(0)
            self.generator_global_rot_3d_pos_angle.append(glob)
            self.generator_3d_pos_angle.append(ang)
        self.generator_3d_pos_angle = np.array(self.generator_3d_pos_angle).reshape(-1, 33)
        jit = np.zeros((N, 8))
        for frame in range(N):                       # second loop of the reference: one jitter draw per frame
            if args.bone_len_scaler == 'different':
                jit[frame] = rnd.randint(-200, 200, size=(8)) / 1000.0
            elif args.bone_len_scaler == 'same':
                jit[frame] = np.repeat(rnd.randint(-200, 200, size=(1)), 8) / 1000.0
            elif args.bone_len_scaler != '':
                raise ValueError("args.bone_len_scaler")
        bl = np.asarray(self.generator_bone_len, dtype=np.float64)
        col = [0, 0, 1, 1, 2, 2, 3, -1, 4, 4, 5, 5, 6, 6, 7]
        for i, c in enumerate(col):
            if c >= 0:
                bl[:, i] = bl[:, i] * (1 + jit[:, c])
        a37 = np.zeros((N, 37), dtype=np.float32)
        a37[:, :33] = self.generator_3d_pos_angle
        a37[:, 34:37] = np.asarray(self.generator_global_rot_3d_pos_angle)
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device=self.device)
        pos = ops.fk_forward(t(a37), t(bl), t(np.asarray(self.generator_root)), 32).cpu().numpy().astype(np.float32)
        return pos, self.generator_3d_pos_angle, self.generator_global_rot_3d_pos_angle, self.generator_bone_len, \
            self.generator_root

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
