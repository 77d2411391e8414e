"""Model classes per reference sub-project (each of which calls its own package ``models``):

  ``models`` itself          lumina_next_t2i/models            NextDiT, NextDiT_2B_patch2, NextDiT_2B_GQA_patch2
  ``models.imagenet``        Next-DiT-ImageNet/models          DiT_Llama, DiT_Llama_600M_patch2, ... (class-conditional)
  ``models.flag_dit``        lumina_t2i/models                 DiT_Llama, DiT_Llama_5B_patch2 (Flag-DiT)
  ``models.moe``             Next-DiT-MoE/models (models2.py)  DiT_Llama, DiT_Llama_600M_patch2_Both (time + space MoE)
  ``models.compositional``   lumina_next_compositional_generation/models   NextDiT with regional text cross-attention
"""
from . import compositional, flag_dit, imagenet, moe  # noqa: F401
from .model import NextDiT, NextDiT_2B_GQA_patch2, NextDiT_2B_patch2  # noqa: F401
