from . import trainer  # noqa: F401
from .train_loop import AMPTrainer, HookBase, SimpleTrainer, TrainerBase  # noqa: F401
