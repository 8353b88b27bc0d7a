"""Harness-only loader for the *reference* DH-AUG Python sources (this container only).

The reference lives read-only at /root/reference/DH-AUG_master and is imported from
there, never copied.  Its hot-path modules drag in GUI / plotting / logging packages
at import time that are unused by the arithmetic; those are
replaced by empty stub modules so the import succeeds on a headless CPU box.

Used by tests/golden/make_golden.py only.  Nothing under tests/ (other than the
generator), bench.py or the package may import this at run time: /root/reference
does not exist on the GPU box.
"""
import argparse
import os
import sys
import types

REF_ROOT = os.environ.get("DHAUG_REFERENCE_ROOT", "/root/reference/DH-AUG_master")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import matplotlib
    matplotlib.use("Agg")
    matplotlib.use = lambda *a, **k: None          # reference calls matplotlib.use("Qt5Agg")
    _stub("matplotlib.backends.backend_qt5agg", FigureCanvasQTAgg=object)
    for name in ("loguru", "h5py", "cv2", "cdflib"):
        _stub(name)
    _stub("thop", profile=lambda *a, **k: None)

    class _Writer:                                    # tensorboardX.SummaryWriter no-op
        def __init__(self, *a, **k):
            self.scalars = []

        def add_scalar(self, name, value, step=None):
            self.scalars.append((name, float(value), step))

        def close(self):
            pass

    _stub("tensorboardX", SummaryWriter=_Writer)
    return _Writer


def make_args(**over):
    """argparse.Namespace with the attributes the hot path reads (R/function_aug/config.py:5-195)."""
    d = dict(
        batch_size=1024, random_seed=0, GAN_OUTPUT_DIM=35, GAN_LAMBDA=10,
        GAN_whether_use_preAngle=True, Gen_DenseDim=256, Dis_DenseDim_3D=256, Dis_DenseDim_2D=256,
        video_Dis_DenseDim_3D=1000, video_Dis_DenseDim_2D=1000,
        GAN_3d_loss_weight=1.0, GAN_2d_loss_weight=0.2,
        GAN_3d_motion_loss_weight=1.0, GAN_2d_motion_loss_weight=1.0,
        bone_len_scaler="different", whether_use_RT=True, flip_GAN_model_input=True,
        GAN_video_playback_input=True, single_or_multi_train_mode="single", architecture="3,3,3",
        record_all_picture=False, motion_Dis_whether_use_3dPos_branch=True,
        motion_Dis_whether_use_3dDiff_branch=True, warmup=2, checkpoint="/tmp/dhaug_ref_ckpt",
        num_workers=0, single_dis_warmup_epoch=4,
    )
    d.update(over)
    return argparse.Namespace(**d)


def load_reference():
    """Returns a dict of the reference modules on the hot path."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    writer_cls = install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    cwd = os.getcwd()
    os.chdir(REF_ROOT)  # some reference modules do sys.path.append(os.getcwd())
    try:
        from models_Fk_GAN import forward_kinematics_DH_model as fkm
        from models_Fk_GAN import Fk_generator as gen
        from models_Fk_GAN import Fk_discriminator as dis
        from models_Fk_GAN import special_operate as sop
        from models_Fk_GAN import model_fk_gan_train as train
        from common import camera, quaternion, h36m_dataset
    finally:
        os.chdir(cwd)
    return dict(fkm=fkm, gen=gen, dis=dis, sop=sop, train=train, camera=camera,
                quaternion=quaternion, h36m=h36m_dataset, Writer=writer_cls)
