from .dense_retrieval_model import DRModel, DRModelForInference, DROutput
from .linear import LinearHead
