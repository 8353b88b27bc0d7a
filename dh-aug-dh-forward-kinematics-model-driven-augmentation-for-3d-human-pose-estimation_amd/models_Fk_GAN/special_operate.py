"""Hot-path symbols of R/models_Fk_GAN/special_operate.py: myResNet (:490-510), Fk_get_boneVecByPose3d (:513-539)."""
import torch
import torch.nn as nn

from .. import autograd_ops as A

# (parent, child) joints of the 15 bones, FK bone order (R/models_Fk_GAN/forward_kinematics_DH_model.py:46-49)
BONE_PARENT = [5, 2, 4, 1, 0, 0, 0, 7, 8, 8, 10, 13, 11, 14, 8]
BONE_CHILD = [6, 3, 5, 2, 4, 1, 7, 8, 10, 13, 11, 14, 12, 15, 9]


class myResNet(nn.Module):
    """relu(fc2(relu(fc1(x))) + x); state_dict keys fc1.{weight,bias}, fc2.{weight,bias} as in the reference.
    Both layers run as bf16 MFMA GEMMs with bias / ReLU / residual fused into the epilogue."""

    def __init__(self, DIM):
        super().__init__()
        self.fc1 = nn.Linear(DIM, DIM)
        self.fc2 = nn.Linear(DIM, DIM)

    def forward(self, input, prec="bf16"):
        return A.res_block(input, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, A.ACT_RELU, 0.0, prec)


def Fk_get_boneVecByPose3d(x, num_joints=16):
    """(N,16,3) -> (N,15,3) child - parent.  The reference multiplies by a dense +-1 (16x15) matrix repeated N
    times; its two call sites (bone lengths, KCS) are fused HIP kernels here (dhaug_bone_length, dhaug_kcs_*), so
    this accessor is plain index arithmetic kept for API completeness."""
    p = torch.as_tensor(BONE_PARENT, device=x.device)
    c = torch.as_tensor(BONE_CHILD, device=x.device)
    return x[:, c] - x[:, p]
