from .dense_retriever import Retriever, SuccessiveRetriever
