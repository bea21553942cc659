"""devqa_amd -- MI355X-native edit-then-evaluate hot path of DE-VQA (BLIP-2 + FT_VL).

Host side is Python mirroring the reference's plugin API; all device arithmetic
goes through the C-ABI library ``csrc/libdevqa_hip.so`` (see include/devqa.h).
"""
__version__ = "0.1.0"
