from .dense_trainer import DRTrainer, GCDenseTrainer
