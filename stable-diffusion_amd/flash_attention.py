"""Names kept for callers of the reference's flash_attention.py (:25, :105).  The Triton kernels
(customtriton.py) are gone; both classes are the MFMA flash-attention modules of attention.py."""
from .attention import CrossAttention as FlashCrossAttention
from .attention import SelfAttention as FlashSelfAttention

__all__ = ["FlashSelfAttention", "FlashCrossAttention"]
