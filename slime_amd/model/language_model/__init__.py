from .llama_attention import HipLlamaAttention, replace_llama_attn_with_hip_attn  # noqa: F401
