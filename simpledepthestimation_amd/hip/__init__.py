"""ctypes binding of libsde_hip.so + torch.autograd wrappers (tensor memory / streams are torch plumbing)."""
