from .beir_dataset import BEIRCorpusDataset, BEIRDataset, BEIRQueryDataset
from .data_collator import DRInferenceCollator, PairCollator, QPCollator, RRInferenceCollator
from .inference_dataset import InferenceDataset, JsonlDataset, TsvDataset
from .train_dataset import DREvalDataset, DRTrainDataset, RREvalDataset, RRTrainDataset
