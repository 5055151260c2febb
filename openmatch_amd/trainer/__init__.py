from .dense_trainer import DRTrainer, GCDenseTrainer
from .reranker_trainer import RRTrainer
