"""Edge-aware smoothness (reference: detectron2/modeling/losses/smoothness_loss.py:L42-80) on the HIP path."""
from ...hip import photometric as HP


def smoothness_loss(depth, image, reversed=False):
    if reversed:
        raise NotImplementedError("reversed=True only flips gradient signs before abs(); unused by the reference")
    return HP.smoothness_loss(depth, image)
