"""supir_amd: MI355X (gfx950) native implementation of SUPIR's restoration-guided EDM sampling hot path.

Host side is Python on PyTorch-ROCm (allocator, streams, RNG, torch.distributed); all heavy compute goes through the
C ABI of libsupir_hip.so (include/supir_hip.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
