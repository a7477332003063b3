"""ultra_pytorch_amd — MI355X-native (gfx950) hot path for unbiased learning to rank.

Drop-in for the `ultra.learning_algorithm.*` / `ultra.ranking_model.DNN` plugin seam of
ULTR-Community/ULTRA_pytorch: the classes here keep the reference's constructor /
`train(input_feed)` / `validation(input_feed)` / `build(input_list)` contracts and are
selected purely by class-path strings in the settings JSON, while the DNN forward /
backward, the listwise and pairwise losses, clipping and the optimizer run as
hand-written HIP kernels behind the C ABI in include/ultr_hip.h.

There is NO CPU fallback: importing the compute modules without the built library, or
calling them without a GPU, raises.
"""
__version__ = "0.1.0"
