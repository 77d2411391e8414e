from .model import NextDiT, NextDiT_2B_GQA_patch2, NextDiT_2B_patch2  # noqa: F401
