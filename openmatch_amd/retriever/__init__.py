from .dense_retriever import Retriever, SuccessiveRetriever
from .reranker import Reranker, RRPredictDataset
