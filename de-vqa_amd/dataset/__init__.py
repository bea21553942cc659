"""Dataset base classes (R/dataset/__init__.py:118-126)."""
from abc import ABC, abstractmethod


class BaseEditData(ABC):
    def __init__(self, data) -> None:
        super().__init__()
        self.data = data

    @abstractmethod
    def dataset_name(self):
        """return dataset name"""
        raise
