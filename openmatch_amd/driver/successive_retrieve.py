"""`retrieve` over corpus partitions searched one after the other and merged
(reference: driver/successive_retrieve.py:17-74)."""
from ..retriever import SuccessiveRetriever
from .retrieve import run


def main():
    run(SuccessiveRetriever)


if __name__ == "__main__":
    main()
