"""ToTensor (reference: detectron2/data/preprocess/formating.py:L8-21): HWC uint8 -> CHW float32 in [0, 1] for img / img_orig and the context
lists (torchvision.transforms.ToTensor's rule for uint8 arrays: permute, float, divide by 255)."""
import numpy as np
import torch

from .build import PREPROCESS_REGISTRY, Preprocess


def to_tensor(img):
    a = np.ascontiguousarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
    return t.to(torch.float32).div(255) if a.dtype == np.uint8 else t.to(torch.float32)


@PREPROCESS_REGISTRY.register()
class ToTensor(Preprocess):
    def forward(self, data_dict):
        for key in data_dict:
            if key in ("img", "img_orig"):
                data_dict[key] = to_tensor(data_dict[key])
            elif key in ("ctx_img", "ctx_img_orig"):
                data_dict[key] = [to_tensor(a) for a in data_dict[key]]
        return data_dict
