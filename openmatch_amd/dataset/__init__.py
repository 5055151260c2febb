from .data_collator import DRInferenceCollator, QPCollator, RRInferenceCollator
