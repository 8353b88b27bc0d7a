"""Drop-in for R/function_aug/dataloader_update.py: random_bl_aug (:18-40) and dataloader_update (:43-107).
The bone algebra of R/utils/gan_utils.py (unit bone vectors, pose rebuild) is one HIP kernel
(dhaug_bone_length_swap); the re-projection uses dhaug_project_to_2d.  The 5 x 15 template table is stored as data
(common/bl_templates.json) instead of being re-read from disk for every batch."""
import json
import os

import numpy as np
import torch

from .. import ops

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "common", "bl_templates.json")) as _f:
    BL_TEMPLATES = np.asarray(json.load(_f)["templates"], dtype=np.float32)        # (5, 15), PoseAug bone order


def random_bl_aug(x, template_idx=None, templates=None):
    """x (N,16,3) -> (N,16,3): bone directions kept, lengths swapped for a random training subject's template.
    template_idx (N,) may be injected; otherwise drawn with np.random.choice as in the reference."""
    templates = BL_TEMPLATES if templates is None else np.asarray(templates, dtype=np.float32)
    if template_idx is None:
        template_idx = np.random.choice(templates.shape[0], x.shape[0])
    lens = torch.as_tensor(templates[np.asarray(template_idx)], device=x.device)
    return ops.bone_length_swap(x.reshape(-1, 16, 3), lens)


class TensorLoader:
    """minimal shuffled batch iterator over device tensors (stands in for DataLoader(PoseDataSet|PoseTarget))"""

    def __init__(self, tensors, batch_size, extras=None):
        self.tensors, self.batch_size, self.extras = tensors, batch_size, extras

    def __len__(self):
        return (self.tensors[0].shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.tensors[0].shape[0]
        perm = torch.randperm(n, device=self.tensors[0].device)
        for i in range(0, n, self.batch_size):
            j = perm[i:i + self.batch_size]
            out = [t[j] for t in self.tensors]
            if self.extras is not None:
                out.insert(2, [self.extras[k] for k in j.tolist()])
            yield out[0] if len(out) == 1 else tuple(out)


def dataloader_update(args, data_dict, device):
    """bone-length swap of the real training poses + re-projection; rebuilds train_gt2d3d_loader / target_3d_loader /
    target_2d_loader (device-resident)."""
    p3, p2, acts, cams = [], [], [], []
    for targets_3d, _, action, cam_param in data_dict['train_gt2d3d_loader']:
        targets_3d, cam_param = targets_3d.to(device), cam_param.to(device)
        targets_3d = random_bl_aug(targets_3d)
        p3.append(targets_3d)
        p2.append(ops.project_to_2d(targets_3d, cam_param))
        acts += list(action)
        cams.append(cam_param)
    p3, p2, cams = torch.cat(p3), torch.cat(p2), torch.cat(cams)
    data_dict['train_gt2d3d_loader'] = TensorLoader([p3, p2, cams], args.batch_size, extras=acts)
    data_dict['target_3d_loader'] = TensorLoader([p3], args.batch_size)
    data_dict['target_2d_loader'] = TensorLoader([p2], args.batch_size)
    return
