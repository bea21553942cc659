"""Editor plugin ABCs: same API as R/editor/vllm_editors/base.py:20-63 (VLLMBaseEditor).

VLLMBaseEditorWithTraining (base.py:67-268: TensorBoard-logged training loop for hyper-network
editors) belongs to the MEND_VL row and is not built in this round.
"""
from abc import ABC, abstractmethod
from typing import Dict, List, Tuple

from ..vllms_for_edit.base import BaseVLLMForEdit


class VLLMBaseEditor(ABC):
    def __init__(self, vllm: BaseVLLMForEdit, device="cuda"):
        if not isinstance(vllm, BaseVLLMForEdit):
            raise RuntimeError("vllm must be a BaseVLLMForEdit")
        self.vllm = vllm
        self.vllm.set_device(device)
        self.device = device if device != "auto" else "cuda:0"
        assert self.if_model_decoder_only()  # only decoder-only llms are supported (base.py:26)

    def if_model_decoder_only(self) -> bool:
        return not self.vllm.model.config.is_encoder_decoder

    @abstractmethod
    def name_of_editor_and_model(self) -> Tuple[str, str]:
        """-> (editor_name, model_name)"""

    @abstractmethod
    def restore_to_original_model(self):
        """restore the original weights after editing"""

    @abstractmethod
    def edit_one_piece(self, request: Dict):
        """request = {'image': path|None, 'prompt': str, 'target_new': str, ...}"""

    @abstractmethod
    def edit_batch(self, requests: List[Dict]):
        """list of requests"""

    @abstractmethod
    def if_can_batch_edit(self) -> bool:
        pass
