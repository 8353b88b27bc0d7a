"""dhaug_amd -- MI355X-native implementation of the DH-AUG hot path (DH forward kinematics + MLP GAN step).

Python here is host plumbing only (device memory, streams, torch.distributed); every arithmetic step of the
path runs in hand-written HIP kernels behind the C-ABI of include/dhaug.h (libdhaug.so, gfx950).  There is no
CPU or PyTorch-eager fallback: calling an op without the library or without a GPU raises.

Module layout mirrors the reference's import paths for the hot path:
    dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model.Forward_Kinematics_DH_Model
    dhaug_amd.models_Fk_GAN.Fk_generator.{Fk_Generator, Video_Fk_Generator}
    dhaug_amd.models_Fk_GAN.Fk_discriminator.{Fk_3D_Discriminator, Fk_2D_Discriminator, calc_gradient_penalty, ...}
    dhaug_amd.models_Fk_GAN.model_fk_gan_train.{my_get_poseFk_model, train_Fk_discriminator, GAN_solutions_FK_generator}
    dhaug_amd.function_aug.config.get_parse_args
"""
__version__ = "0.1.0"

from . import _lib            # noqa: F401  (ctypes binding; the .so is loaded on first use)
