"""tg_hip -- Python binding of libterragan_hip.so (include/terragan_hip.h) and the explicit
forward/backward engines built on it.  There is NO CPU or eager-PyTorch fallback: importing
`tg_hip.lib` without the built library raises, and every op requires CUDA(HIP) tensors."""
