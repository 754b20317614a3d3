"""gar_amd — MI355X-native Grasp-Any-Region region-captioning hot path (host side).

Python here is orchestration only (weights, preprocessing, sequencing, data-parallel sharding); every device
computation goes through the C-ABI HIP library ``libgar_hip.so`` (include/gar_hip.h). There is no CPU or
PyTorch fallback: importing :mod:`gar_amd.hip` without the built library raises.
"""
from .configuration_gar import GARConfig, PerceptionLMConfig, TextConfig, VisionConfig  # noqa: F401

__all__ = ["GARConfig", "PerceptionLMConfig", "TextConfig", "VisionConfig"]
