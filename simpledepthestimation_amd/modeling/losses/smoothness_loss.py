"""Edge-aware smoothness (reference: detectron2/modeling/losses/smoothness_loss.py:L42-80) on the HIP path."""
from ...hip import photometric as HP


def smoothness_loss(depth, image, reversed=False):
    """mean|d_x(1/d normalised)| e^{-mean_c|d_x I|} + the same in y.  `reversed` flips the sign of every finite difference (gradient_x / gradient_y,
    smoothness_loss.py:L9-39) -- of the depth gradients AND of the image gradients, both of which only enter through abs() -- so the loss and its
    gradient are the same either way; the kernel computes the forward-difference form."""
    del reversed
    return HP.smoothness_loss(depth, image)
