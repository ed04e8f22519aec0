"""realpdebench_amd -- MI355X-native backend for RealPDEBench's FNO3d train step and autoregressive rollout.

Host code mirrors the reference's plugin surface for this path (``load_model`` -> ``Model`` protocol:
``forward`` / ``train_loss`` / ``load_checkpoint`` / reference-named ``state_dict``); the math runs in
hand-written HIP kernels behind the C ABI of ``include/rpb.h`` (``csrc/librpb_hip.so``).  There is no
CPU or eager-PyTorch fallback: calling an op without the built library or without a GPU raises.
"""
__version__ = "0.1.0"
