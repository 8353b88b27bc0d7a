"""Flag surface of R/function_aug/config.py:5-195 (names, types, defaults preserved) for the hot path's callers.
Table-driven restatement; booleans parse as in the reference (str(x).lower() == 'true')."""
import argparse

_bool = lambda x: (str(x).lower() == 'true')

# (flag, default, type)
_FLAGS = [
    ("dataset", "h36m", str), ("keypoints", "gt", str), ("actions", "*", str), ("checkpoint", "checkpoint/debug", str),
    ("snapshot", 2, int), ("note", "debug", str), ("evaluate", "", str), ("resume", "", str),
    ("posenet_name", "videopose", str), ("stages", 4, int), ("dropout", 0.25, float),
    ("batch_size", 1024, int), ("epochs", 50, int), ("decay_epoch", 0, int),
    ("lr_g", 1.0e-4, float), ("lr_d", 1.0e-4, float), ("lr_p", 1.0e-4, float),
    ("random_seed", 0, int), ("downsample", 1, int), ("pretrain", False, _bool), ("s1only", False, _bool),
    ("s1s5only", False, _bool), ("num_workers", 0, int), ("warmup", 2, int), ("df", 2, int),
    ("data_enhancement_method", "GAN", str), ("generator_whole_number", 10000, int),
    ("generator_choose_BoneLen", True, _bool), ("bone_len_scaler", "different", str),
    ("generator_choose_root_pos", True, _bool), ("generator_global_rot", True, _bool),
    ("GAN_OUTPUT_DIM", 32 + 3, int), ("GAN_LAMBDA", 10, int), ("GAN_whether_use_preAngle", True, _bool),
    ("motion_Dis_whether_use_3dPos_branch", True, _bool), ("motion_Dis_whether_use_3dDiff_branch", True, _bool),
    ("Dis_DenseDim_3D", 1000, int), ("Dis_DenseDim_2D", 1000, int), ("Gen_DenseDim", 1000, int),
    ("video_Dis_DenseDim_3D", 1000, int), ("video_Dis_DenseDim_2D", 1000, int),
    ("GAN_3d_loss_weight", 1, float), ("GAN_2d_loss_weight", 0.2, float),
    ("GAN_3d_motion_loss_weight", 1, float), ("GAN_2d_motion_loss_weight", 1, float),
    ("GAN_whether_rand_root", True, _bool), ("set_demo_mode", False, _bool), ("GAN_checkpoint", "checkpoint", str),
    ("GAN_resume", "", str), ("record_all_picture", True, _bool), ("additional_train_epoch", 60, int),
    ("additional_LR_decay", 0.95, float), ("single_dis_warmup_epoch", 4, int), ("video_over_200mm", False, _bool),
    ("whether_use_RT", True, _bool), ("flip_pos_model_input", True, _bool), ("flip_GAN_model_input", True, _bool),
    ("Pos_video_playback_input", True, _bool), ("GAN_video_playback_input", True, _bool), ("gpu_id", "0", str),
    ("Path_3DPW", "3DPW_dataSet", str), ("single_or_multi_train_mode", "single", str), ("architecture", "3,3,3", str),
]


def get_parse_args(argv=None):
    parser = argparse.ArgumentParser(description='DH-AUG hot-path flags')
    for name, default, typ in _FLAGS:
        parser.add_argument('--' + name, default=default, type=typ)
    parser.add_argument('--no_max', dest='max_norm', action='store_false')
    parser.set_defaults(max_norm=True)
    args = parser.parse_args(argv)
    if args.resume and args.evaluate:
        raise SystemExit('Invalid flags: --resume and --evaluate cannot be set at the same time')
    return args


def synth_args(batch_size, D=256, **over):
    """argparse.Namespace with the attributes the hot path reads, for synthetic runs (bench.py, smoke, tools): the
    reference's defaults except the dense widths (D instead of 1000 for the single-frame networks)"""
    d = dict(batch_size=batch_size, random_seed=0, GAN_OUTPUT_DIM=35, GAN_LAMBDA=10, GAN_whether_use_preAngle=True,
             Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D, video_Dis_DenseDim_3D=1000, video_Dis_DenseDim_2D=1000,
             GAN_3d_loss_weight=1.0, GAN_2d_loss_weight=0.2, GAN_3d_motion_loss_weight=1.0, GAN_2d_motion_loss_weight=1.0,
             bone_len_scaler="different", whether_use_RT=True, flip_GAN_model_input=True, GAN_video_playback_input=True,
             single_or_multi_train_mode="single", architecture="3,3,3", motion_Dis_whether_use_3dPos_branch=True,
             motion_Dis_whether_use_3dDiff_branch=True, warmup=2, num_workers=0, single_dis_warmup_epoch=4)
    d.update(over)
    return argparse.Namespace(**d)
